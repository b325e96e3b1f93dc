// shard_group.hpp -- the shards of ONE proof running concurrently in this process, one host thread each (zkr_multi.hip
// run_sharded; zkr_prove.hip calc_h_split).  Plain C++ (no HIP): host_arith_shim.cpp compiles it for the CPU tests.
#pragma once
#include <condition_variable>
#include <mutex>
#include <vector>

namespace zkr {

// The shards of ONE proof running concurrently in this process (zkr_multi.hip run_sharded), one host thread each: what a split
// calcH needs from the others -- where their vectors are, and a barrier.  A shard that fails aborts the group: everybody waiting
// (now or later) returns false instead of waiting for a thread that will never arrive.
struct ShardGroup {
  struct Vecs { void *va, *vb, *ca, *cb, *dh; };  // Fr vectors of a proof slot (zkr_internal.hpp ProofSlot), as the owner sees them
  unsigned parts = 0;
  int klog = 0;                       // parts = 2^klog
  std::vector<Vecs> vecs;             // [part], published by the part's thread before the first barrier
  bool split_h = false;               // every precondition of the split holds (zkr_multi.hip run_sharded)
  bool solo = false;                  // measurement only (zkr_bench_shard_split_solo): ONE shard runs its phases with its own buffers
                                      // standing in for the others' -- the time of a shard alone on its GPU, the result meaningless
  double phase_ms[8][8] = {};         // [part][phase]: host time enqueue -> stream idle of the split's phases (bench / tests)
  std::mutex mu;
  std::condition_variable cv;
  unsigned waiting = 0, generation = 0;
  bool failed = false;
  bool barrier() {                    // false: the group was aborted
    std::unique_lock<std::mutex> lk(mu);
    if (failed) return false;
    if (solo) return true;
    const unsigned gen = generation;
    if (++waiting == parts) { waiting = 0; generation++; cv.notify_all(); return true; }
    cv.wait(lk, [&] { return generation != gen || failed; });
    return generation != gen;         // everybody arrived (an abort after that is seen at the next barrier)
  }
  void abort() {
    std::lock_guard<std::mutex> lk(mu);
    failed = true;
    cv.notify_all();
  }
};

}  // namespace zkr
