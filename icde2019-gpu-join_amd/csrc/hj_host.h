// hj_host.h — host-only helpers shared by hj_api.hip and hj_host.cpp (no HIP types).
#ifndef HJ_HOST_H_
#define HJ_HOST_H_

#include <stdint.h>

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

// NUMA topology from sysfs (no libnuma): number of nodes with memory, and the CPUs of one node that this process may run on
int host_numa_nodes();
std::vector<int> host_node_cpus(int node);

} // namespace hj
#endif
