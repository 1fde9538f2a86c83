// Prime-factor DCT for the 2^k+1 grid lengths (pfa.hip).
#pragma once
#include "common.h"

namespace dotsocp {

struct PfaPlan;

bool pfa_supported(i64 n);
PfaPlan *pfa_plan_create(i64 n);        // nullptr when n is not one of the supported lengths (or on allocation failure)
void pfa_plan_destroy(PfaPlan *p);

// fused t-axis solve: lambda(line, k) = cy[G % nyE] + cx[G / nyE] + ct[k], G = line0 + row * gRow + (line index in the row)
struct PfaSolveArgs {
    double kscale;
    const double *cy, *cx, *ct;
    i64 nyE, line0, gRow;
};

// Strided axis: `nrows` rows of `nyLines` lines; line (row, y) starts at row * srow + y (dst: drow), element k a
// further k * sel (del) on.  mode 0: DCT-II, 1: DCT-III, 2: DCT-II, division by kscale * lambda, DCT-III (sargs).
// src == dst is allowed.
int pfa_launch_strided(const PfaPlan *p, const double *src, double *dst, i64 nyLines, i64 nrows, i64 srow, i64 sel,
                       i64 drow, i64 del, int mode, const PfaSolveArgs *sargs, hipStream_t st);
// Axis 0: line L starts at L * sline (dst: dline), elements contiguous.
int pfa_launch_axis0(const PfaPlan *p, const double *src, double *dst, i64 nLines, i64 sline, i64 dline, int inverse,
                     hipStream_t st);

}  // namespace dotsocp
