// hj_host.h — host-only helpers shared by hj_api.hip and hj_host.cpp (no HIP types).
#ifndef HJ_HOST_H_
#define HJ_HOST_H_

#include <stdint.h>

#include <vector>

namespace hj {

// shard of a key for the level-0 splits (host mirror of the device function digit_of<1>)
uint32_t host_shard_of(int32_t key, uint32_t nshards);

// Level-0 split of (K, Pv) into `parts` contiguous runs of (oK, oP) on `threads` host threads; off[parts+1].
// false: a host thread could not be started.
bool host_level0_split(const int32_t *K, const int32_t *Pv, uint64_t n, uint32_t parts, uint32_t threads,
                       int32_t *oK, int32_t *oP, std::vector<uint64_t> &off);

} // namespace hj
#endif
