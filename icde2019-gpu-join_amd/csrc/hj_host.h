// hj_host.h — host-only helpers shared by hj_api.hip and hj_host.cpp (no HIP types).
#ifndef HJ_HOST_H_
#define HJ_HOST_H_

#include <stdint.h>

#include <atomic>
#include <functional>
#include <memory>
#include <vector>

namespace hj {

// shard of a key for the level-0 splits (host mirror of the device function digit_of<1>)
uint32_t host_shard_of(int32_t key, uint32_t nshards);

// Level-0 split of (K, Pv) into `parts` contiguous runs of (oK, oP) on `threads` host threads; off[parts+1].
// false: a host thread could not be started.  pin_cpus (optional): the workers bind themselves to these CPUs — the socket of
// the NUMA node the staging buffers live on (the reference binds its partitioning threads to sockets the same way,
// partition-primitives.cu:139,197,210,227).
bool host_level0_split(const int32_t *K, const int32_t *Pv, uint64_t n, uint32_t parts, uint32_t threads,
                       int32_t *oK, int32_t *oP, std::vector<uint64_t> &off, const std::vector<int> *pin_cpus = nullptr);

// The same split in ONE pass over the input, for the co-processing path: a partition comes out as a list of blocks (each worker takes
// blocks of host_split_block_size() tuples (2^8 … 2^20) from its own arena of the staging columns as its partitions fill up: nothing can overflow,
// there is no histogram pass and no partition-id column).  oK / oP must hold host_split_blocks_capacity(n, parts, threads) tuples and
// be 64-byte aligned; oP is written only when Pv is given.  blocks: sorted by (partition, start); part_size[parts].
struct HostBlock { uint32_t part; uint64_t start, count; };
uint32_t host_split_workers(uint64_t n, uint32_t threads);                      // workers the split really starts: one per 2^16 tuples at least
uint32_t host_split_block_size(uint64_t n, uint32_t parts, uint32_t workers);    // (workers = host_split_workers(n, threads))
uint64_t host_split_blocks_capacity(uint64_t n, uint32_t parts, uint32_t threads);
// While the workers run: worker t's arena is [arena[t], arena[t + 1]) and everything in [arena[t], done(t)) is complete — whole blocks,
// all of them full, their streaming stores fenced.  A caller that does not care which partition a tuple belongs to (one residency group)
// uploads those ranges while the split is still running.
struct HostSplitProgress {
    std::vector<uint64_t> arena;                    // [threads + 1]
    std::unique_ptr<std::atomic<uint64_t>[]> upto;  // one counter per worker, a cache line apart
    uint64_t done(uint32_t t) const { return upto[(size_t)t * 8].load(std::memory_order_acquire); }
};
// while_running (optional): called again and again by the CALLING thread (which then takes no share of the split) until every worker has
// finished, and once more after that.
bool host_level0_split_blocks(const int32_t *K, const int32_t *Pv, uint64_t n, uint32_t parts, uint32_t threads, int32_t *oK, int32_t *oP,
                              std::vector<HostBlock> &blocks, std::vector<uint64_t> &part_size, const std::vector<int> *pin_cpus = nullptr,
                              const std::function<void(const HostSplitProgress &)> *while_running = nullptr);

// NUMA topology from sysfs (no libnuma): number of nodes with memory, and the CPUs of one node that this process may run on
int host_numa_nodes();
std::vector<int> host_node_cpus(int node);

} // namespace hj
#endif
