// device_math.hpp -- per-pixel model of the RS differential-SfM solve, shared by all kernels.
//
// Compiled with -ffp-contract=off: the operation order below is the reference's
// (nonlinearRefinement.cc:32-52 for the residual, minimal.cc:255-275 for the scoring error) so that
// integer outputs (inlier masks / counts, LM decisions) can be compared bit-exactly with the CPU oracle.
#pragma once

#include <hip/hip_runtime.h>

#include "rsdsfm_internal.hpp"

namespace rsdsfm {

__device__ __forceinline__ double clampd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }

// LevenbergMarquardtStrategy::StepAccepted (Ceres 1.14 levenberg_marquardt_strategy.cc)
__host__ __device__ __forceinline__ double radius_accept(double radius, double q) {
    double t = 2.0 * q - 1.0;
    double f = 1.0 - t * t * t;
    if (f < 1.0 / 3.0) f = 1.0 / 3.0;
    radius = radius / f;
    if (radius > kMaxRadius) radius = kMaxRadius;
    return radius;
}

// Per-pixel constants of the dense depth problem for a fixed pose (v, w, k):
//   r(rho) = u - pred(rho),  pred_j = beta * -1 * (rho * a_j + t1_j - t2_j + t3_j)      (nonlinearRefinement.cc:36-49)
//   J = d r / d rho = beta * a
struct PixelModel {
    double nbeta;       // beta * -1.0
    double a0, a1;      // x v_z - v_x ,  y v_z - v_y
    double t01, t02, t03, t11, t12, t13;
    double ux, uy;
    double J0, J1;

    __device__ __forceinline__ void init(double x, double y, double ux_, double uy_, double alpha, double alpha_k,
                                         const Pose& p, double two_over) {
        double beta = two_over * (alpha + p.k * alpha_k);  // (2/(2+k)) * (alpha + k alpha_k)
        nbeta = beta * -1.0;
        a0 = x * p.v[2] - p.v[0];
        a1 = y * p.v[2] - p.v[1];
        t01 = x * y * p.w[0];
        t02 = (1.0 + x * x) * p.w[1];
        t03 = y * p.w[2];
        t11 = (1.0 + y * y) * p.w[0];
        t12 = x * y * p.w[1];
        t13 = x * p.w[2];
        ux = ux_;
        uy = uy_;
        J0 = beta * a0;
        J1 = beta * a1;
    }
    __device__ __forceinline__ void residual(double rho, double& r0, double& r1) const {
        double p0 = nbeta * (rho * a0 + t01 - t02 + t03);
        double p1 = nbeta * (rho * a1 + t11 - t12 - t13);
        r0 = ux - p0;
        r1 = uy - p1;
    }
};

// minimal.cc:255-270: residual norm of the flow predicted from (v, w, k, rho)
__device__ __forceinline__ double point_error(double x, double y, double ux, double uy, double alpha, double alpha_k,
                                              const Pose& p, double two_over, double rho) {
    double beta = (alpha + p.k * alpha_k) * two_over;
    double av0 = p.v[0] + (-x) * p.v[2];
    double av1 = p.v[1] + (-y) * p.v[2];
    double bw0 = (-x * y) * p.w[0] + (1 + x * x) * p.w[1] + (-y) * p.w[2];
    double bw1 = (-(1 + y * y)) * p.w[0] + (x * y) * p.w[1] + x * p.w[2];
    double e0 = beta * (av0 * rho + bw0) - ux;
    double e1 = beta * (av1 * rho + bw1) - uy;
    return sqrt(e0 * e0 + e1 * e1);
}

// ---- wave64 / workgroup reductions (deterministic: fixed butterfly order) ----
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    return v;
}

// Arrival counter: returns true in exactly one workgroup per launch -- the last one whose wave 0 called it.
// Payload written by wave 0 with agent-scope relaxed atomic stores (write-through) BEFORE the call is
// visible to the last arriver through agent-scope relaxed atomic loads (cdna guide G16, "8-B agent
// atomics both sides").  Hierarchical (8 group counters + 1 top counter) so that no word sees more than
// nblocks/8 returning atomics.  Counters are zero at entry and are reset by their last arriver.
// Must be called by all threads of the workgroup (contains __syncthreads()).
__device__ __forceinline__ bool arrive_last(unsigned* tickets, int nblocks) {
    __shared__ int s_last;
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int group = blockIdx.x & 7;
        const int ngroup = (nblocks - group + 7) >> 3;
        const int ngroups = nblocks < 8 ? nblocks : 8;
        int last = 0;
        unsigned t = __hip_atomic_fetch_add(&tickets[group], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == (unsigned)(ngroup - 1)) {
            __hip_atomic_store(&tickets[group], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned tt = __hip_atomic_fetch_add(&tickets[8], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tt == (unsigned)(ngroups - 1)) {
                __hip_atomic_store(&tickets[8], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last = 1;
            }
        }
        s_last = last;
    }
    __syncthreads();
    return s_last != 0;
}

__device__ __forceinline__ void store_agent(double* p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double load_agent(const double* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace rsdsfm
