/*
 * aad_launch.h - kernel launches that can carry a completion event on their own dispatch packet.
 *
 * AADHip_ContextSignalNextRun (include/aad_hip.h) asks for events to be recorded when a plan run's work starts / is done.  Every
 * plan run is ONE kernel, so the events ride on that kernel's dispatch (hipExtLaunchKernelGGL's start / stop event) instead of on
 * barrier packets of their own around it: measured on MI355X (tools/microbench/ubench_event_gap.hip,
 * profiles/r03_microbench_event_gap.txt) a hipEventRecord behind every kernel of a back-to-back sequence costs the queue
 * 2.9 us per launch, the attached event nothing.  The events travel from the entry point to the launch site in a
 * thread-local (the launch helpers are templates several levels down); the first AAD_LAUNCH of the thread takes them.
 */
#ifndef AAD_LAUNCH_H_INCLUDED
#define AAD_LAUNCH_H_INCLUDED

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

namespace aad {
struct LaunchSignal {
  hipEvent_t start, stop; /* either may be null */
};
extern thread_local LaunchSignal tl_launch_signal; /* defined in aad_hip_engine.hip; both null when no run asked for events */
}

#define AAD_LAUNCH(kernel, grid, block, lds, stream, ...)                                                           \
  do {                                                                                                              \
    const aad::LaunchSignal aad_signal_ = aad::tl_launch_signal;                                                    \
    aad::tl_launch_signal = aad::LaunchSignal{nullptr, nullptr};                                                    \
    if (aad_signal_.start != nullptr || aad_signal_.stop != nullptr)                                                \
      hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, aad_signal_.start, aad_signal_.stop, 0, __VA_ARGS__); \
    else                                                                                                            \
      hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                            \
  } while (0)

#endif /* AAD_LAUNCH_H_INCLUDED */
