"""ASAN + UBSan over the CPU-side C of the repository: oracle/mex_kernels.c, the seven MEX gateways
and the libmx stand-in, driven by tests/san/san_driver.c -- the operators on small and degenerate grids, and every
argument-error path of the gateways (none of which needs a device).  GPU AddressSanitizer is not available on the
pool; the device side is covered by the guard bands of tests/test_gpu_canary.py."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MEX = os.path.join(ROOT, "dot-socp_amd", "mex")
OUT = os.path.join(ROOT, "tests", "san", "_build")
GATES = ["mexProjSoc", "mexBFd", "mexBFdConj", "mexBFd1d", "mexBFdConj1d", "dotsocp_inpalm_mex", "dotsocp_level_mex"]


def test_cpu_side_c_is_clean_under_asan_and_ubsan():
    os.makedirs(OUT, exist_ok=True)
    san = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-g", "-O1"]
    inc = ["-I" + os.path.join(MEX, "compile_check"), "-I" + os.path.join(ROOT, "include"), "-I" + MEX]
    objs = []
    for g in GATES:
        o = os.path.join(OUT, g + ".o")
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra"] + san + inc + ["-DmexFunction=mexFunction_" + g, "-c",
                               os.path.join(MEX, g + ".c"), "-o", o])
        objs.append(o)
    for src in (os.path.join(ROOT, "tests", "fake_mx", "fake_mx.c"), os.path.join(ROOT, "oracle", "mex_kernels.c"),
                os.path.join(ROOT, "tests", "san", "san_driver.c")):
        o = os.path.join(OUT, os.path.basename(src)[:-2] + ".o")
        subprocess.check_call(["gcc", "-std=c99", "-Wall"] + san + inc + ["-c", src, "-o", o])
        objs.append(o)
    libdir = os.path.join(ROOT, "dot-socp_amd", "lib")
    exe = os.path.join(OUT, "san_driver")
    subprocess.check_call(["gcc"] + san + objs + ["-o", exe, "-L" + libdir, "-ldotsocp", "-Wl,-rpath," + libdir, "-lm"])
    # leak checking stays on for this executable's own allocations; what the (uninstrumented) HIP runtime keeps for the
    # life of the process is none of this test's business
    supp = os.path.join(OUT, "lsan.supp")
    with open(supp, "w") as f:
        f.write("leak:libamdhip64\nleak:libhsa-runtime64\nleak:libdotsocp\nleak:libamd_comgr\nleak:librocprofiler\n")
    env = dict(os.environ, ASAN_OPTIONS="abort_on_error=0:detect_leaks=1:halt_on_error=1",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1", LSAN_OPTIONS="suppressions=" + supp)
    run = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-4000:]
    assert "0 failed" in run.stdout
    assert "ERROR: AddressSanitizer" not in run.stderr and "runtime error" not in run.stderr


def test_slab_thread_layer_is_clean_under_tsan():
    """ThreadSanitizer over the slab-thread layer (dot-socp_amd/csrc/defer.h, defer.hip: one issuing thread per time slab
    in the single-process multi-slab modes): built with g++ against a stub of the few HIP declarations it uses and
    driven by tests/san/defer_tsan.cpp with the call pattern of the time-slab loop -- per-stream order kept, every stream
    wait bound to the record that preceded it in host order, no data race, no deadlock."""
    os.makedirs(OUT, exist_ok=True)
    csrc = os.path.join(ROOT, "dot-socp_amd", "csrc")
    exe = os.path.join(OUT, "defer_tsan")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-pthread", "-x", "c++",
                           "-I" + os.path.join(ROOT, "tests", "san", "hipstub"), "-I" + csrc,
                           os.path.join(csrc, "defer.hip"), os.path.join(ROOT, "tests", "san", "defer_tsan.cpp"), "-o", exe])
    run = subprocess.run([exe], env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1"), capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-4000:]
    assert "0 failed" in run.stdout and "ThreadSanitizer" not in run.stderr


def test_host_side_helpers_of_the_download_path_under_tsan_and_asan():
    """dot-socp_amd/csrc/hostmem.hip (first touch of download targets and the sigma scaling of alpha / beta on several host
    threads) is plain C++: built with g++ under ThreadSanitizer and under AddressSanitizer + UBSan, driven by
    tests/san/hostmem_check.cpp -- contents untouched at every alignment, scaling equal to the serial loop bit for bit."""
    os.makedirs(OUT, exist_ok=True)
    csrc = os.path.join(ROOT, "dot-socp_amd", "csrc")
    for name, flags in (("tsan", ["-fsanitize=thread"]), ("asan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"])):
        exe = os.path.join(OUT, "hostmem_" + name)
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-pthread"] + flags + ["-x", "c++", "-I" + csrc,
                               os.path.join(csrc, "hostmem.hip"), os.path.join(ROOT, "tests", "san", "hostmem_check.cpp"), "-o", exe])
        for threads in ("1", "5"):
            run = subprocess.run([exe], env=dict(os.environ, DOTSOCP_HOST_COPY_THREADS=threads, TSAN_OPTIONS="halt_on_error=1",
                                                 ASAN_OPTIONS="halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1"),
                                 capture_output=True, text=True, timeout=600)
            assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-4000:]
            assert f"threads {threads}, 0 failed" in run.stdout and "Sanitizer" not in run.stderr
