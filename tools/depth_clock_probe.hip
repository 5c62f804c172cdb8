// depth_clock_probe.hip -- why the memory and the arithmetic of the dense depth solve's streaming kernel do not overlap.
//
// depth_lm_batch_kernel (csrc/depth_kernels.hip, BASELINE configs[1]) runs 4 x 1280x720 solves per launch.  Alone, its memory
// traffic takes ~40 us and its arithmetic ~35 us; together they take ~45 us, and neither more waves per SIMD, nor a register prefetch
// of the next operands, nor an LDS-DMA ring changes that (round 4).  This tool shows what does decide it: the SHADER CLOCK.  It runs
// three kernels on the library's own per-pixel code (csrc/lm_common.hpp, included by path) over the same 4 x 921 600 pixels --
//   full    : the library kernel's loop (6 x 16-byte non-temporal loads, 2 pixels through 3 speculated LM iterations, one store)
//   memory  : the same loads and store, a handful of adds
//   compute : the same arithmetic on register operands, no memory traffic
// -- and lets one lane of every 97th workgroup read s_memtime (shader clocks) and s_memrealtime (100 MHz) around its life: the
// quotient is the clock that workgroup ran at.  Kernel durations come from the dispatches' own timestamps (hipExtLaunchKernelGGL
// start / stop events), i.e. what rocprofv3 --kernel-trace reports.  Measured on MI355X: the full kernel runs at ~1.8 GHz, the
// skeletons at 2.2-2.3 GHz: fp64 issue at full rate plus ~4.5 TB/s of HBM traffic exceeds the power budget, the chip clocks down,
// and the arithmetic -- 35 us at 2.23 GHz -- becomes ~43 us at 1.82 GHz: the kernel is bound by fp64 issue AT THE THROTTLED CLOCK.
//
// Standalone:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I rs-aware-differential-sfm_amd/csrc -I include tools/depth_clock_probe.hip -o tools/depth_clock_probe
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "lm_common.hpp"

using namespace rsdsfm;

#define CHECK(x)                                                                  \
    do {                                                                          \
        hipError_t e = (x);                                                       \
        if (e != hipSuccess) {                                                    \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

typedef double d2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 nt_load(const double2* p) {
    d2v v = __builtin_nontemporal_load(reinterpret_cast<const d2v*>(p));
    return make_double2(v.x, v.y);
}
__device__ __forceinline__ void nt_store(double2* p, double2 o) {
    d2v v;
    v.x = o.x;
    v.y = o.y;
    __builtin_nontemporal_store(v, reinterpret_cast<d2v*>(p));
}

struct Item {
    const double2 *q, *u, *a2, *ak2;
    double2* rho2;
};
struct Args {
    Item item[4];
    int64_t n;
    Pose pose;
    unsigned long long* probes;  // [grid.y][grid.x][2] = {shader clocks, 100 MHz ticks} of the workgroup's life (every 97th workgroup)
    double* sink;
};

// MODE 0 = full, 1 = memory only, 2 = compute only
template <int MODE>
__global__ __launch_bounds__(256) void probe_kernel(Args args) {
    const unsigned long long clk0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
    const Item& it = args.item[blockIdx.y];
    const int tid = threadIdx.x;
    LmPlanFirst pf;
    pf.write_which = 2;
    {
        double r = kInitialRadius;
        for (int j = 0; j < KMAX; ++j) {
            pf.inv_cand[j] = 1.0 / r;
            r = radius_accept(r, 1.0);
        }
    }
    const Pose pose = args.pose;
    const double two_over = 2.0 / (2.0 + pose.k);
    double acc[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) acc[s] = 0.0;
    const int64_t npairs = args.n >> 1;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    if (MODE == 0) {
        for (int64_t p = (int64_t)blockIdx.x * blockDim.x + tid; p < npairs; p += stride) {
            const double2 qa = nt_load(it.q + 2 * p), qb = nt_load(it.q + 2 * p + 1);
            const double2 ua = nt_load(it.u + 2 * p), ub = nt_load(it.u + 2 * p + 1);
            const double2 al = nt_load(it.a2 + p), ak = nt_load(it.ak2 + p);
            double2 out;
            out.x = lm_pixel(qa.x, qa.y, ua.x, ua.y, al.x, ak.x, pose, two_over, pf, acc);
            out.y = lm_pixel(qb.x, qb.y, ub.x, ub.y, al.y, ak.y, pose, two_over, pf, acc);
            nt_store(it.rho2 + p, out);
        }
    } else if (MODE == 1) {
        for (int64_t p = (int64_t)blockIdx.x * blockDim.x + tid; p < npairs; p += stride) {
            const double2 qa = nt_load(it.q + 2 * p), qb = nt_load(it.q + 2 * p + 1);
            const double2 ua = nt_load(it.u + 2 * p), ub = nt_load(it.u + 2 * p + 1);
            const double2 al = nt_load(it.a2 + p), ak = nt_load(it.ak2 + p);
            double2 out;
            out.x = qa.x + qa.y + ua.x + ua.y + al.x + ak.x;
            out.y = qb.x + qb.y + ub.x + ub.y + al.y + ak.y;
            acc[0] += out.x + out.y;
            nt_store(it.rho2 + p, out);
        }
    } else {
        double2 qa = make_double2(1e-3 * tid, 2e-3 * tid), qb = make_double2(1.5e-3 * tid, 1e-3 * tid);
        double2 ua = make_double2(1e-4 * tid, 2e-4), ub = make_double2(3e-4, 1e-4 * tid);
        double2 out = make_double2(0.0, 0.0);
        for (int64_t p = (int64_t)blockIdx.x * blockDim.x + tid; p < npairs; p += stride) {
            out.x = lm_pixel(qa.x, qa.y, ua.x, ua.y, 1.0, 0.5, pose, two_over, pf, acc);
            out.y = lm_pixel(qb.x, qb.y, ub.x, ub.y, 1.0, 0.5, pose, two_over, pf, acc);
            qa.x += out.x * 1e-9, qb.y += out.y * 1e-9;
        }
        if (tid == 0) nt_store(it.rho2 + blockIdx.x, out);
    }
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < NS; ++k) s += acc[k];
    s = wave_sum(s);
    if ((tid & 63) == 0 && s == 1.2345e-300) args.sink[0] = s;  // keeps the sums alive
    if (tid == 0 && blockIdx.x % 97 == 0) {
        unsigned long long* pr = args.probes + 2 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x);
        pr[0] = __builtin_amdgcn_s_memtime() - clk0;
        pr[1] = __builtin_amdgcn_s_memrealtime() - rt0;
    }
}

int main(int argc, char** argv) {
    const int64_t n = 1280 * 720;
    const int pairs = 4, sets = 6, reps = argc > 1 ? atoi(argv[1]) : 60;  // 6 rotating buffer sets x 4 pairs x 52 MB: past the 256 MiB Infinity Cache
    const int grid_x = 300;
    std::vector<double> hq(2 * n), hu(2 * n), ha(n), hak(n);
    for (int64_t i = 0; i < n; ++i) {
        const double x = ((i / 720) - 633.0) / 995.0, y = ((i % 720) - 370.0) / 994.0;
        hq[2 * i] = x, hq[2 * i + 1] = y;
        hu[2 * i] = 0.02 + 0.01 * x, hu[2 * i + 1] = 0.02 - 0.01 * y;
        ha[i] = 1.0 + 0.001 * y, hak[i] = 0.5 + 0.2 * y;
    }
    struct Set {
        double *q, *u, *a, *ak, *rho;
    };
    std::vector<Set> S(sets * pairs);
    for (auto& s : S) {
        CHECK(hipMalloc(&s.q, 16 * n));
        CHECK(hipMalloc(&s.u, 16 * n));
        CHECK(hipMalloc(&s.a, 8 * n));
        CHECK(hipMalloc(&s.ak, 8 * n));
        CHECK(hipMalloc(&s.rho, 8 * n));
        CHECK(hipMemcpy(s.q, hq.data(), 16 * n, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(s.u, hu.data(), 16 * n, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(s.a, ha.data(), 8 * n, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(s.ak, hak.data(), 8 * n, hipMemcpyHostToDevice));
    }
    const size_t nprobe = (size_t)2 * pairs * grid_x;
    unsigned long long* d_probes;  // one probe block per launch: the launches of a kernel run back to back (the clock is a property of the sustained load)
    double* d_sink;
    CHECK(hipMalloc(&d_probes, 8 * nprobe * reps));
    CHECK(hipMalloc(&d_sink, 64));
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    std::vector<hipEvent_t> e0(reps), e1(reps);
    for (int r = 0; r < reps; ++r) {
        CHECK(hipEventCreate(&e0[r]));
        CHECK(hipEventCreate(&e1[r]));
    }
    Pose pose = {{0.70710678, 0.70710678, 0.0}, {0.0, 0.0, 0.0087}, 0.0};
    const char* names[3] = {"full", "memory-only", "compute-only"};
    printf("# depth_clock_probe: %d back-to-back launches per kernel of 4 x 1280x720 pixels, grid %d x %d x 256 threads; duration = the dispatch's own timestamps\n", reps, grid_x, pairs);
    printf("%-13s %9s %9s %9s | shader clock of the probed workgroups (MHz): %6s %6s %6s | workgroup life (shader clocks) %8s\n", "kernel", "avg us", "med us", "min us", "mean",
           "min", "max", "mean");
    for (int round = 0; round < 2; ++round)
        for (int mode = 0; mode < 3; ++mode) {
            CHECK(hipMemsetAsync(d_probes, 0, 8 * nprobe * reps, st));
            for (int r = 0; r < reps; ++r) {
                Args a;
                for (int i = 0; i < pairs; ++i) {
                    const Set& s = S[(r % sets) * pairs + i];
                    a.item[i] = {(const double2*)s.q, (const double2*)s.u, (const double2*)s.a, (const double2*)s.ak, (double2*)s.rho};
                }
                a.n = n, a.pose = pose, a.probes = d_probes + nprobe * r, a.sink = d_sink;
                if (mode == 0) hipExtLaunchKernelGGL(probe_kernel<0>, dim3(grid_x, pairs), dim3(256), 0, st, e0[r], e1[r], 0, a);
                if (mode == 1) hipExtLaunchKernelGGL(probe_kernel<1>, dim3(grid_x, pairs), dim3(256), 0, st, e0[r], e1[r], 0, a);
                if (mode == 2) hipExtLaunchKernelGGL(probe_kernel<2>, dim3(grid_x, pairs), dim3(256), 0, st, e0[r], e1[r], 0, a);
                CHECK(hipGetLastError());
            }
            CHECK(hipStreamSynchronize(st));
            std::vector<float> ms;
            std::vector<double> mhz, life;
            std::vector<unsigned long long> h(nprobe * reps);
            CHECK(hipMemcpy(h.data(), d_probes, 8 * nprobe * reps, hipMemcpyDeviceToHost));
            for (int r = reps / 4; r < reps; ++r) {
                float f = 0.f;
                CHECK(hipEventElapsedTime(&f, e0[r], e1[r]));
                ms.push_back(f);
                for (size_t k = 0; k < nprobe; k += 2)
                    if (h[nprobe * r + k + 1]) mhz.push_back(100.0 * (double)h[nprobe * r + k] / (double)h[nprobe * r + k + 1]), life.push_back((double)h[nprobe * r + k]);
            }
            std::sort(ms.begin(), ms.end());
            double avg = 0, cm = 0, lm = 0;
            for (float f : ms) avg += f;
            for (double c : mhz) cm += c;
            for (double c : life) lm += c;
            if (round == 1)
                printf("%-13s %9.2f %9.2f %9.2f | %52s %6.0f %6.0f %6.0f | %38.0f\n", names[mode], 1e3 * avg / ms.size(), 1e3 * ms[ms.size() / 2], 1e3 * ms[0], "", cm / mhz.size(),
                       *std::min_element(mhz.begin(), mhz.end()), *std::max_element(mhz.begin(), mhz.end()), lm / life.size());
        }
    return 0;
}
