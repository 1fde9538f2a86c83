// Micro-benchmark: achievable HBM bandwidth of the access patterns used by the loop kernels
// (tools only; not part of the library).  hipcc --offload-arch=gfx950 -O3 tools/bw_probe.hip -o bw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef long long i64;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void copy8(const double *a, double *b, i64 n) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) b[i] = a[i] * 1.0001;
}
__global__ void copy16(const double2 *a, double2 *b, i64 n2) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (i64)gridDim.x * blockDim.x) {
        double2 v = a[i]; v.x *= 1.0001; v.y *= 1.0001; b[i] = v;
    }
}
// 10-plane SoA, tile 64(y) x XB(x), marching over t: like k_cone_fused (8 B per lane)
template <int XB>
__global__ void __launch_bounds__(64 * XB) soa8(const double *in, double *out, i64 ny, i64 nx, i64 nt) {
    const i64 y = (i64)blockIdx.x * 64 + threadIdx.x, x = (i64)blockIdx.y * XB + threadIdx.y;
    const i64 Nz = ny * nx * nt;
    for (i64 t = 0; t < nt; ++t) {
        const i64 i = y + ny * (x + nx * t);
        double v[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) v[j] = in[j * Nz + i];
#pragma unroll
        for (int j = 0; j < 10; ++j) out[j * Nz + i] = v[j] * 1.0001;
    }
}
// same, two consecutive y per lane (16 B per lane): tile 128(y) x XB(x)
template <int XB>
__global__ void __launch_bounds__(64 * XB) soa16(const double *in, double *out, i64 ny, i64 nx, i64 nt) {
    const i64 y = ((i64)blockIdx.x * 64 + threadIdx.x) * 2, x = (i64)blockIdx.y * XB + threadIdx.y;
    const i64 Nz = ny * nx * nt;
    for (i64 t = 0; t < nt; ++t) {
        const i64 i = y + ny * (x + nx * t);
        double2 v[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) v[j] = *(const double2 *)(in + j * Nz + i);
#pragma unroll
        for (int j = 0; j < 10; ++j) { v[j].x *= 1.0001; v[j].y *= 1.0001; *(double2 *)(out + j * Nz + i) = v[j]; }
    }
}
// same as soa8 but a thread block walks t in chunks given by gridDim.z
template <int XB>
__global__ void __launch_bounds__(64 * XB) soa8c(const double *in, double *out, i64 ny, i64 nx, i64 nt, i64 tc) {
    const i64 y = (i64)blockIdx.x * 64 + threadIdx.x, x = (i64)blockIdx.y * XB + threadIdx.y;
    const i64 Nz = ny * nx * nt;
    const i64 t0 = blockIdx.z * tc, t1 = (t0 + tc < nt) ? t0 + tc : nt;
    for (i64 t = t0; t < t1; ++t) {
        const i64 i = y + ny * (x + nx * t);
        double v[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) v[j] = in[j * Nz + i];
#pragma unroll
        for (int j = 0; j < 10; ++j) out[j * Nz + i] = v[j] * 1.0001;
    }
}

// soa8 with the next step's loads issued BEFORE the current step's stores (register double buffer):
// vmcnt counts loads and stores in issue order, so a load issued after a store cannot be waited for
// without also waiting for the store's completion
template <int XB>
__global__ void __launch_bounds__(64 * XB) soa8p(const double *in, double *out, i64 ny, i64 nx, i64 nt) {
    const i64 y = (i64)blockIdx.x * 64 + threadIdx.x, x = (i64)blockIdx.y * XB + threadIdx.y;
    const i64 Nz = ny * nx * nt;
    double cur[10], nxt[10];
    i64 i = y + ny * x;
#pragma unroll
    for (int j = 0; j < 10; ++j) cur[j] = in[j * Nz + i];
    for (i64 t = 0; t < nt; ++t) {
        const i64 in1 = i + ny * nx;
        if (t + 1 < nt) {
#pragma unroll
            for (int j = 0; j < 10; ++j) nxt[j] = in[j * Nz + in1];
        }
#pragma unroll
        for (int j = 0; j < 10; ++j) out[j * Nz + i] = cur[j] * 1.0001;
#pragma unroll
        for (int j = 0; j < 10; ++j) cur[j] = nxt[j];
        i = in1;
    }
}

// tile TY (y) x 1 (x): TY consecutive y of one column per step = TY*8 contiguous bytes per plane
template <int TY>
__global__ void __launch_bounds__(TY) soa8y(const double *in, double *out, i64 ny, i64 nx, i64 nt, i64 tc) {
    const i64 y = (i64)blockIdx.x * TY + threadIdx.x, x = blockIdx.y;
    const i64 Nz = ny * nx * nt;
    const i64 t0 = blockIdx.z * tc, t1 = (t0 + tc < nt) ? t0 + tc : nt;
    for (i64 t = t0; t < t1; ++t) {
        const i64 i = y + ny * (x + nx * t);
        double v[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) v[j] = in[j * Nz + i];
#pragma unroll
        for (int j = 0; j < 10; ++j) out[j * Nz + i] = v[j] * 1.0001;
    }
}
// march along x instead of t: consecutive steps are ny*8 bytes apart (same DRAM neighbourhood)
__global__ void __launch_bounds__(256) soa8x(const double *in, double *out, i64 ny, i64 nx, i64 nt, i64 xc) {
    const i64 y = (i64)blockIdx.x * 256 + threadIdx.x, t = blockIdx.z;
    const i64 Nz = ny * nx * nt;
    const i64 x0 = blockIdx.y * xc, x1 = (x0 + xc < nx) ? x0 + xc : nx;
    for (i64 x = x0; x < x1; ++x) {
        const i64 i = y + ny * (x + nx * t);
        double v[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) v[j] = in[j * Nz + i];
#pragma unroll
        for (int j = 0; j < 10; ++j) out[j * Nz + i] = v[j] * 1.0001;
    }
}

// blocked layout (AoSoA): the ten planes of 64 consecutive cells are contiguous (5 KB)
template <int XB>
__global__ void __launch_bounds__(64 * XB) aosoa8(const double *in, double *out, i64 ny, i64 nx, i64 nt, i64 tc) {
    const i64 y = (i64)blockIdx.x * 64 + threadIdx.x, x = (i64)blockIdx.y * XB + threadIdx.y;
    const i64 t0 = blockIdx.z * tc, t1 = (t0 + tc < nt) ? t0 + tc : nt;
    for (i64 t = t0; t < t1; ++t) {
        const i64 i = y + ny * (x + nx * t);
        const i64 base = (i >> 6) * 640 + (i & 63);
        double v[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) v[j] = in[base + j * 64];
#pragma unroll
        for (int j = 0; j < 10; ++j) out[base + j * 64] = v[j] * 1.0001;
    }
}

// soa8c as a persistent kernel: `gridDim.x` workgroups walk the (tile, chunk) units in dispatch order, so that the
// resident workgroups stay on neighbouring tiles of the same layers (a ragged dispatch spreads them over many layers)
template <int XB>
__global__ void __launch_bounds__(64 * XB) soa8pers(const double *in, double *out, i64 ny, i64 nx, i64 nt, i64 tc) {
    const i64 nyb = ny / 64, nxb = nx / XB, nzb = (nt + tc - 1) / tc;
    const i64 Nz = ny * nx * nt;
    for (i64 u = blockIdx.x; u < nyb * nxb * nzb; u += gridDim.x) {
        const i64 by = u % nyb, bx = (u / nyb) % nxb, bz = u / (nyb * nxb);
        const i64 y = by * 64 + threadIdx.x, x = bx * XB + threadIdx.y;
        const i64 t0 = bz * tc, t1 = (t0 + tc < nt) ? t0 + tc : nt;
        for (i64 t = t0; t < t1; ++t) {
            const i64 i = y + ny * (x + nx * t);
            double v[10];
#pragma unroll
            for (int j = 0; j < 10; ++j) v[j] = in[j * Nz + i];
#pragma unroll
            for (int j = 0; j < 10; ++j) out[j * Nz + i] = v[j] * 1.0001;
        }
    }
}

// soa8c with a plane stride that differs from the plane length (ps doubles between the planes of in / out, `skew` more
// between in and out): do the ten streams of a wave collide on HBM channels when the planes are a multiple of 8 MiB apart?
template <int XB>
__global__ void __launch_bounds__(64 * XB) soa8s(const double *in, double *out, i64 ny, i64 nx, i64 nt, i64 tc, i64 ps) {
    const i64 y = (i64)blockIdx.x * 64 + threadIdx.x, x = (i64)blockIdx.y * XB + threadIdx.y;
    const i64 t0 = blockIdx.z * tc, t1 = (t0 + tc < nt) ? t0 + tc : nt;
    for (i64 t = t0; t < t1; ++t) {
        const i64 i = y + ny * (x + nx * t);
        double v[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) v[j] = in[j * ps + i];
#pragma unroll
        for (int j = 0; j < 10; ++j) out[j * ps + i] = v[j] * 1.0001;
    }
}

// soa8c fed by LDS-DMA: the ten planes of a step arrive in an LDS ring D steps deep (global_load_lds_dwordx4, no
// registers held by loads in flight), the lanes read their values from LDS, scale and store.  Does a marching tile
// get closer to the copy rate when two or three steps of loads are always in flight?
__device__ __forceinline__ void glds16p(const void *gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int D>
__global__ void __launch_bounds__(256) soa8dma(const double *in, double *out, i64 ny, i64 nx, i64 nt, i64 tc) {
    extern __shared__ double ring[];                 // [D][10][4][64]
    const int lane = threadIdx.x, xl = threadIdx.y, tid = xl * 64 + lane;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), l64 = tid & 63;
    const i64 y0 = (i64)blockIdx.x * 64, x0 = (i64)blockIdx.y * 4;
    const i64 Nz = ny * nx * nt;
    const i64 t0 = blockIdx.z * tc, t1 = (t0 + tc < nt) ? t0 + tc : nt;
    const unsigned base = (unsigned)(uintptr_t)ring;
    // piece q = plane * 2 + half (20 per step, 5 per wave): lanes 0..31 column 2 half, lanes 32..63 column 2 half + 1
    auto dma = [&](i64 t, int slot) {
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int q = wave * 5 + i, plane = q >> 1, half = q & 1;
            const i64 col = x0 + 2 * half + (l64 >> 5);
            const double *g = in + plane * Nz + y0 + ny * (col + nx * t) + 2 * (l64 & 31);
            glds16p(g, base + (unsigned)(((slot * 10 + plane) * 4 + 2 * half) * 64) * 8u);
        }
    };
    for (int d = 0; d < D - 1; ++d)
        if (t0 + d < t1) dma(t0 + d, d);
    for (i64 t = t0; t < t1; ++t) {
        const int slot = (int)((t - t0) % D);
        if (t + D - 1 < t1) dma(t + D - 1, (int)((t - t0 + D - 1) % D));
        // DMA(t) done when at most the younger operations are outstanding (steady state): see the text above
        if (t + D - 1 < t1 && t - t0 >= D - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"i"((D - 1) * 5 + (D - 1) * 10 > 63 ? 63 : (D - 1) * 5 + (D - 1) * 10) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const i64 i = y0 + lane + ny * (x0 + xl + nx * t);
        double v[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) v[j] = ring[((slot * 10 + j) * 4 + xl) * 64 + lane];
#pragma unroll
        for (int j = 0; j < 10; ++j) out[j * Nz + i] = v[j] * 1.0001;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // slot read by everyone before it is refilled
    }
}

template <class F>
static double timeit(F f, int reps = 5) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main() {
    const i64 ny = 1024, nx = 1024, nt = 127, Nz = ny * nx * nt, N = 10 * Nz;
    double *a, *b;
    const i64 SLACK = 10 * (1 << 20);   // room for padded plane strides
    CK(hipMalloc(&a, (N + SLACK) * 8)); CK(hipMalloc(&b, (N + SLACK) * 8));
    CK(hipMemset(a, 0, (N + SLACK) * 8)); CK(hipMemset(b, 0, (N + SLACK) * 8));
    const double gb = 2.0 * N * 8 / 1e9;
    double ms;
    ms = timeit([&] { hipLaunchKernelGGL(copy8, dim3(16384), dim3(256), 0, 0, a, b, N); });
    printf("copy8  grid-stride 16384 blocks: %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
    ms = timeit([&] { hipLaunchKernelGGL(copy8, dim3((unsigned)(N / 256)), dim3(256), 0, 0, a, b, N); });
    printf("copy8  one elem/thread        : %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
    ms = timeit([&] { hipLaunchKernelGGL(copy16, dim3(16384), dim3(256), 0, 0, (const double2 *)a, (double2 *)b, N / 2); });
    printf("copy16 grid-stride 16384 blocks: %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
    ms = timeit([&] { hipLaunchKernelGGL(copy16, dim3((unsigned)(N / 512)), dim3(256), 0, 0, (const double2 *)a, (double2 *)b, N / 2); });
    printf("copy16 one elem/thread        : %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
    ms = timeit([&] { hipLaunchKernelGGL(soa8<4>, dim3(ny / 64, nx / 4), dim3(64, 4), 0, 0, a, b, ny, nx, nt); });
    printf("soa8  64x4 tile, march all t  : %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
    ms = timeit([&] { hipLaunchKernelGGL(soa8<8>, dim3(ny / 64, nx / 8), dim3(64, 8), 0, 0, a, b, ny, nx, nt); });
    printf("soa8  64x8 tile, march all t  : %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
    ms = timeit([&] { hipLaunchKernelGGL(soa8c<4>, dim3(ny / 64, nx / 4, 8), dim3(64, 4), 0, 0, a, b, ny, nx, nt, (i64)16); });
    printf("soa8  64x4 tile, 8 chunks     : %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
    ms = timeit([&] { hipLaunchKernelGGL(soa8c<4>, dim3(ny / 64, nx / 4, 127), dim3(64, 4), 0, 0, a, b, ny, nx, nt, (i64)1); });
    printf("soa8  64x4 tile, 1 cell/thread: %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
    for (i64 pad : {(i64)0, (i64)16, (i64)32, (i64)64, (i64)272, (i64)528, (i64)2064, (i64)4112, (i64)65552, (i64)(1 << 19) + 528}) {
        ms = timeit([&] { hipLaunchKernelGGL(soa8s<4>, dim3(ny / 64, nx / 4, 8), dim3(64, 4), 0, 0, a, b, ny, nx, nt, (i64)16, Nz + pad); });
        printf("soa8s 64x4 tile, 8 chunks, plane stride Nz + %7lld doubles: %.3f ms  %.0f GB/s\n", pad, ms, gb / ms * 1e3);
    }
    for (i64 skew : {(i64)16, (i64)528, (i64)4112}) {
        ms = timeit([&] { hipLaunchKernelGGL(soa8s<4>, dim3(ny / 64, nx / 4, 8), dim3(64, 4), 0, 0, a, b + skew, ny, nx, nt, (i64)16, Nz); });
        printf("soa8s 64x4 tile, 8 chunks, out shifted by %5lld doubles    : %.3f ms  %.0f GB/s\n", skew, ms, gb / ms * 1e3);
    }
    for (unsigned wgs : {1024u, 2048u, 4096u}) {
        ms = timeit([&] { hipLaunchKernelGGL(soa8pers<4>, dim3(wgs), dim3(64, 4), 0, 0, a, b, ny, nx, nt, (i64)16); });
        printf("soa8pers 64x4, 16-layer chunks, %4u persistent workgroups: %.3f ms  %.0f GB/s\n", wgs, ms, gb / ms * 1e3);
    }
    ms = timeit([&] { hipLaunchKernelGGL(soa8pers<4>, dim3(2048), dim3(64, 4), 0, 0, a, b, ny, nx, nt, (i64)4); });
    printf("soa8pers 64x4,  4-layer chunks, 2048 persistent workgroups: %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
    ms = timeit([&] { hipLaunchKernelGGL(soa8pers<4>, dim3(2048), dim3(64, 4), 0, 0, a, b, ny, nx, nt, (i64)127); });
    printf("soa8pers 64x4, whole-t chunks,  2048 persistent workgroups: %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
    hipFuncSetAttribute((const void *)soa8dma<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void *)soa8dma<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void *)soa8dma<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (i64 tcv : {(i64)16, (i64)32, (i64)127}) {
        const unsigned nz = (unsigned)((nt + tcv - 1) / tcv);
        ms = timeit([&] { hipLaunchKernelGGL(soa8dma<2>, dim3(ny / 64, nx / 4, nz), dim3(64, 4), 2 * 20480, 0, a, b, ny, nx, nt, tcv); });
        printf("soa8dma ring 2, chunks of %3lld   : %.3f ms  %.0f GB/s\n", tcv, ms, gb / ms * 1e3);
        ms = timeit([&] { hipLaunchKernelGGL(soa8dma<3>, dim3(ny / 64, nx / 4, nz), dim3(64, 4), 3 * 20480, 0, a, b, ny, nx, nt, tcv); });
        printf("soa8dma ring 3, chunks of %3lld   : %.3f ms  %.0f GB/s\n", tcv, ms, gb / ms * 1e3);
        ms = timeit([&] { hipLaunchKernelGGL(soa8dma<4>, dim3(ny / 64, nx / 4, nz), dim3(64, 4), 4 * 20480, 0, a, b, ny, nx, nt, tcv); });
        printf("soa8dma ring 4, chunks of %3lld   : %.3f ms  %.0f GB/s\n", tcv, ms, gb / ms * 1e3);
    }
    ms = timeit([&] { hipLaunchKernelGGL(soa8p<4>, dim3(ny / 64, nx / 4), dim3(64, 4), 0, 0, a, b, ny, nx, nt); });
    printf("soa8p 64x4 tile, prefetch next: %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
    ms = timeit([&] { hipLaunchKernelGGL(soa8y<256>, dim3(ny / 256, nx, 1), dim3(256), 0, 0, a, b, ny, nx, nt, nt); });
    printf("soa8y 256x1 tile, march all t : %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
    ms = timeit([&] { hipLaunchKernelGGL(soa8y<1024>, dim3(ny / 1024, nx, 1), dim3(1024), 0, 0, a, b, ny, nx, nt, nt); });
    printf("soa8y 1024x1 tile, march all t: %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
    ms = timeit([&] { hipLaunchKernelGGL(soa8y<256>, dim3(ny / 256, nx, 127), dim3(256), 0, 0, a, b, ny, nx, nt, (i64)1); });
    printf("soa8y 256x1 tile, 1 cell/thr  : %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
    ms = timeit([&] { hipLaunchKernelGGL(soa8x, dim3(ny / 256, 1, nt), dim3(256), 0, 0, a, b, ny, nx, nt, nx); });
    printf("soa8x 256y, march all x       : %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
    ms = timeit([&] { hipLaunchKernelGGL(soa8x, dim3(ny / 256, 16, nt), dim3(256), 0, 0, a, b, ny, nx, nt, (i64)64); });
    printf("soa8x 256y, march 64 x        : %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
    ms = timeit([&] { hipLaunchKernelGGL(aosoa8<4>, dim3(ny / 64, nx / 4, 1), dim3(64, 4), 0, 0, a, b, ny, nx, nt, nt); });
    printf("aosoa8 64x4 tile, march all t : %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
    ms = timeit([&] { hipLaunchKernelGGL(aosoa8<4>, dim3(ny / 64, nx / 4, 127), dim3(64, 4), 0, 0, a, b, ny, nx, nt, (i64)1); });
    printf("aosoa8 64x4 tile, 1 cell/thr  : %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
    ms = timeit([&] { hipLaunchKernelGGL(aosoa8<1>, dim3(ny / 64, nx, 1), dim3(64, 1), 0, 0, a, b, ny, nx, nt, nt); });
    printf("aosoa8 64x1 tile, march all t : %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
    ms = timeit([&] { hipLaunchKernelGGL(soa16<4>, dim3(ny / 128, nx / 4), dim3(64, 4), 0, 0, a, b, ny, nx, nt); });
    printf("soa16 128x4 tile, march all t : %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
    ms = timeit([&] { hipLaunchKernelGGL(soa16<2>, dim3(ny / 128, nx / 2), dim3(64, 2), 0, 0, a, b, ny, nx, nt); });
    printf("soa16 128x2 tile, march all t : %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
    return 0;
}
