// Device-side words that cross between QUEUES of one GPU while kernels run (exchange pipelines, csrc/hsr_exec.hip):
// the tail of a K1 launch on the caller's stream publishes a tile's moments for the all-reduce on the side stream, and a
// side-stream kernel publishes the coefficients for a K3 pre-phase on the caller's stream - without an event, because
// every hipEventRecord / hipStreamWaitEvent on the caller's stream costs a ~5 us bubble between two K1 launches.
//
// Protocol (same as the ticket counters of hsr_srf.hip): the payload is stored at agent scope (`sc1`: written through
// the XCD's L2), the writer waits for its stores (s_waitcnt vmcnt(0)), then sets / increments the word with a relaxed
// agent-scope atomic; the reader polls the word with agent-scope loads and reads the payload with agent-scope loads.
// (Release / acquire fences instead - buffer_wbl2 / buffer_inv - flush the whole L2 with the output image's dirty lines
// in it: 40 us per launch, measured in round 2.)
// Every poll has a wall-clock limit (the 100 MHz s_memrealtime counter): a wave that can never be satisfied would hang
// the GPU - past the limit it writes a code to the pipeline's error word, which the host reads at the next drain, and
// goes on.
#pragma once
#include "hsr_common.h"

#define HSR_SYNC_TIMEOUT_S 20

namespace hsr {

__device__ __forceinline__ unsigned long long sync_realtime() {   // constant 100 MHz counter, the same on every XCD
  unsigned long long t;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}

__device__ __forceinline__ unsigned int ld_agent_u32(const unsigned int* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent_u32(unsigned int* p, unsigned int v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Wait until the word has reached `target` (wrap-safe "at least": the words only ever count up).  One thread calls it.
// code: what to leave in *err when the limit is hit (1 = gate of the moments, 2 = coefficients of a K3, 3 = drain gate).
__device__ __forceinline__ void wait_word_at_least(const unsigned int* word, unsigned int target, unsigned int* err, unsigned int code) {
  if ((int)(ld_agent_u32(word) - target) >= 0) return;
  const unsigned long long t0 = sync_realtime();
  for (;;) {
    __builtin_amdgcn_s_sleep(16);                                // ~1 k cycles between polls
    if ((int)(ld_agent_u32(word) - target) >= 0) return;
    if (sync_realtime() - t0 > (unsigned long long)HSR_SYNC_TIMEOUT_S * 100000000ull) {
      if (err) st_agent_u32(err, code);
      return;
    }
  }
}

}  // namespace hsr
