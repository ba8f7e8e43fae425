// fe_impl.h — per-shape dispatch record shared by fe_api.hip and the per-shape translation units
// (one fe_shape_<name>.hip per compiled shape, so that the shapes build in parallel).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>

#include <cstdlib>

#include "fe_kernels.hip.h"
#include "fe_frame8.hip.h"
#include "tb_kernels.hip.h"

namespace fe {

struct Impl {
    int C1, NL, C2, F2, KB, NFFT, HOP, KT, LOW, FR, TA, LN, BD;
    const tb::TbImpl* tb;   // time-batched engine (tb_kernels.hip.h) or nullptr
    size_t lds_bytes;
    int occ;              // resident workgroups per CU
    bool many_persist;    // companion: also used beyond occ x #CUs streams (persistent workgroups)
    bool wg8;             // the 512-thread per-hop kernel (fe_frame8.hip.h) is built for this shape
    int n_units, u_max;
    bool staged;
    size_t skip_floats;   // per stream, 0 when the skips stay in LDS
    size_t dbg_floats;
    int dbg_stages;
    const PackedOffsets* off;
    void (*launch)(const FrameArgs&, int max_wgs, hipStream_t, hipError_t*);
    void (*launch_pipe)(const FrameArgs&, hipStream_t, hipError_t*);     // time-pipelined offline / spec launch (a.pipe_p workgroups per stream)
    void (*dbg_stage)(int, int*, int*, size_t*);
    const char* name = nullptr;      // the line of fe_shapes.def this record was compiled from (fe_shape.hip.in): part of fe_last_step_kernel's answer
    bool many_one_round = true;      // companion: used from the first stream above the shape's own plan (false: only beyond occ x #CUs streams, as persistent workgroups)
};

// the instantiation a launcher picked, as fe_last_step_kernel reports it
template <class S>
constexpr const char* frame_kernel_name(bool dbg, bool per_hop, bool persist) {
    if (dbg) return S::LOW == 2 ? "fe_frame_kernel<LOW=2, debug>" : S::LOW == 1 ? "fe_frame_kernel<LOW=1, debug>" : "fe_frame_kernel<debug>";
    if (per_hop && !persist) return S::LOW == 2 ? "fe_frame_kernel<LOW=2, per-hop>" : S::LOW == 1 ? "fe_frame_kernel<LOW=1, per-hop>" : "fe_frame_kernel<per-hop>";
    if (per_hop) return S::LOW == 2 ? "fe_frame_kernel<LOW=2, per-hop, persistent>" : S::LOW == 1 ? "fe_frame_kernel<LOW=1, per-hop, persistent>" : "fe_frame_kernel<per-hop, persistent>";
    return S::LOW == 2 ? "fe_frame_kernel<LOW=2, generic>" : S::LOW == 1 ? "fe_frame_kernel<LOW=1, generic>" : "fe_frame_kernel<generic>";
}

template <class S, bool DBG, int MODE, bool T1, bool PERSIST>
void launch_one(const FrameArgs& a, int grid_x, hipStream_t st, hipError_t* err) {
    // the opt-in for > 64 KiB of dynamic LDS is a per-device function attribute: one flag per device, set once
    // (an engine may live on any GPU of the process; relaxed atomics - setting it twice is harmless)
    static std::atomic<bool> attr_set[kMaxDevices];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = 0;
    if (!attr_set[dev].load(std::memory_order_relaxed)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fe_frame_kernel<S, DBG, MODE, T1, PERSIST>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)Lds<S>::BYTES);
        if (e != hipSuccess) { *err = e; return; }
        attr_set[dev].store(true, std::memory_order_relaxed);
    }
    dim3 grid(grid_x), block(kThreads);
    note_kernel(frame_kernel_name<S>(DBG, T1, PERSIST));
    hipLaunchKernelGGL((fe_frame_kernel<S, DBG, MODE, T1, PERSIST>), grid, block, Lds<S>::BYTES, st, a);
    *err = hipGetLastError();
}

// a.step_kernel (fe_set_step_kernel; the handle's default comes from the environment variable FE_WG8, else 1):
//   0 = the four-wave kernel everywhere; 1 = the 512-thread kernel (fe_frame8.hip.h: two waves per SIMD, channel-grouped GRU gates)
//   for the per-hop step of the shapes it is built for, up to one stream per CU; 2 = also above that (persistent workgroups)
template <class S, bool DBG, bool PERSIST>
void launch_one8(const FrameArgs& a, int grid_x, hipStream_t st, hipError_t* err) {
    static std::atomic<bool> attr_set[kMaxDevices];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = 0;
    if (!attr_set[dev].load(std::memory_order_relaxed)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fe_frame8_kernel<S, DBG, PERSIST>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)Wg8<S>::BYTES);
        if (e != hipSuccess) { *err = e; return; }
        attr_set[dev].store(true, std::memory_order_relaxed);
    }
    dim3 grid(grid_x), block(kThreads8);
    note_kernel(DBG ? "fe_frame8_kernel<debug>" : PERSIST ? "fe_frame8_kernel<persistent>" : "fe_frame8_kernel");
    hipLaunchKernelGGL((fe_frame8_kernel<S, DBG, PERSIST>), grid, block, Wg8<S>::BYTES, st, a);
    *err = hipGetLastError();
}

// max_wgs: workgroups that are resident at once (one per CU: 129+ KiB of LDS and waves_per_eu(1,1)); a batch with more
// streams runs on a grid of max_wgs PERSISTENT workgroups, each walking its streams b, b + grid, ...
template <class S>
void launch_impl(const FrameArgs& a, int max_wgs, hipStream_t st, hipError_t* err) {
    const int slots = max_wgs * Lds<S>::OCC;         // resident workgroups
    const int grid = a.B < slots ? a.B : slots;
    if constexpr (Wg8<S>::OK) {
        if (a.step_kernel > 0 && a.mode == FE_MODE_STREAM && a.T == 1) {
            const int grid8 = a.B < max_wgs ? a.B : max_wgs;       // (one 512-thread workgroup per CU)
#ifdef FE_PROBE_HOT
            const bool dbg8 = a.dbg != nullptr;
#else
            const bool dbg8 = a.dbg != nullptr || a.clk != nullptr;
#endif
            if (grid8 == a.B) {
                if (dbg8) launch_one8<S, true, false>(a, grid8, st, err);
                else launch_one8<S, false, false>(a, grid8, st, err);
                return;
            }
            if (a.step_kernel > 1 && !dbg8) { launch_one8<S, false, true>(a, grid8, st, err); return; }
        }
    }
#ifdef FE_PROBE_HOT
    if (a.dbg != nullptr) launch_one<S, true, -1, false, true>(a, grid, st, err);
#else
    if (a.dbg != nullptr || a.clk != nullptr) launch_one<S, true, -1, false, true>(a, grid, st, err);     // fe_debug_step / fe_profile_step
#endif
    else if (a.mode == FE_MODE_STREAM && a.T == 1) {                                   // the per-hop hot path
#ifdef FE_EXP_NONPERSIST      // experiment: one workgroup per stream at any batch (the hardware queues what is not resident); LOW = 1 keeps nothing per workgroup in global memory
        if (S::LOW == 1) { launch_one<S, false, FE_MODE_STREAM, true, false>(a, a.B, st, err); return; }
#endif
        if (grid == a.B) launch_one<S, false, FE_MODE_STREAM, true, false>(a, grid, st, err);
        else launch_one<S, false, FE_MODE_STREAM, true, true>(a, grid, st, err);
    }
    else launch_one<S, false, -1, false, true>(a, grid, st, err);                        // chunked streaming, fe_spec_step, fe_offline
}

// Time-pipelined launch: B * pipe_p workgroups that wait on each other inside the kernel - a cooperative launch, so
// that the runtime guarantees (or refuses) their co-residency instead of a spin-wait deadlock.
template <class S>
void launch_pipe_impl(const FrameArgs& a, hipStream_t st, hipError_t* err) {
    {
    auto* fn = &fe_frame_kernel<S, false, -1, false, true, true>;
    static std::atomic<bool> attr_set[kMaxDevices];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = 0;
    if (!attr_set[dev].load(std::memory_order_relaxed)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)Lds<S>::BYTES);
        if (e != hipSuccess) { *err = e; return; }
        attr_set[dev].store(true, std::memory_order_relaxed);
    }
    FrameArgs args = a;
    void* kargs[] = {&args};
    note_kernel("fe_frame_kernel<time-pipelined>");
    *err = hipLaunchCooperativeKernel(reinterpret_cast<const void*>(fn), dim3(a.B * a.pipe_p), dim3(kThreads), kargs, (unsigned int)Lds<S>::BYTES, st);
    }
}

template <class S>
void dbg_stage_impl(int s, int* rows, int* cols, size_t* off) {
    *rows = DebugLayout<S>::rows(s);
    *cols = DebugLayout<S>::cols(s);
    *off = DebugLayout<S>::offset(s);
}

template <class S>
Impl make_impl() {
    const tb::TbImpl* tbp = nullptr;
    if constexpr (S::TB) {
        static const tb::TbImpl tbi = tb::make_tb_impl<S>();
        tbp = &tbi;
    }
    if constexpr (S::BIDIR) {       // noncausal: no frame-by-frame kernel (the reverse-time scan needs all frames): time-batched engine only
        return Impl{S::C1, S::NL, S::C2, S::F2, S::KB, S::NFFT, S::HOP, S::KT, S::LOW, 0, 0, 0, 1, tbp, (size_t)0, 1, false, false, S::NU, Pack<S>::umax(), false,
                    (size_t)0, DebugLayout<S>::total(), DebugLayout<S>::n_stages, &Pack<S>::v, nullptr, nullptr, &dbg_stage_impl<S>};
    } else {
    Impl im{S::C1, S::NL, S::C2, S::F2, S::KB, S::NFFT, S::HOP, S::KT, S::LOW, S::FRNN ? 1 : 0, S::LB, S::LN ? 1 : 0, 0, tbp, Lds<S>::BYTES, Lds<S>::OCC, Lds<S>::MANY_PERSIST, Wg8<S>::OK, S::NU, Pack<S>::umax(), Lds<S>::STAGED,
            Lds<S>::SKIPS_LDS ? (size_t)0 : (size_t)(S::NL + 1) * S::F1 * S::C1,
            DebugLayout<S>::total(), DebugLayout<S>::n_stages, &Pack<S>::v, &launch_impl<S>, &launch_pipe_impl<S>, &dbg_stage_impl<S>};
    im.many_one_round = Lds<S>::MANY_ONE_ROUND;
    return im;
    }
}


}  // namespace fe
