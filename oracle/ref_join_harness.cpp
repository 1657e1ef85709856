/*
 * ref_join_harness.cpp — ORACLE tooling (test infrastructure): a command-line driver around the
 * REFERENCE's own CPU join, `joinCpu` + `h_hashMurmur`
 * (/root/reference/src/hash_join_clustered_probe.cu:2013-2059), the only code in the reference that
 * computes a join on the CPU.  oracle/Makefile (`ref` target) cuts that span out of the reference file
 * where it lies into oracle/_ref/joincpu_extract.inc at build time (git-ignored, never committed) and
 * this file #includes it next to the reference's own common-host.h (for time_block); no header,
 * library or tool is stood in for.  Nothing here restates reference code.
 *
 * usage:  refjoin R.bin S.bin        (raw int32 relations, the reference's .bin format)
 * stdout: the reference's own prints — "===<build count>", "===<s> <c> <g>", "<g> join results"
 *         where s = number of matching pairs, g = sum of the matching S keys (uint32 wrap).
 * Run with OMP_NUM_THREADS=1: `s` is not in the reduction clause of the probe loop (a data race in the
 * reference).
 */
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <vector>

#include "common-host.h"

#include "_ref/joincpu_extract.inc"

static bool slurp(const char *path, std::vector<int32_t> &v) {
    FILE *fp = fopen(path, "rb");
    if (!fp) return false;
    fseek(fp, 0, SEEK_END);
    long bytes = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    v.resize((size_t)bytes / 4 + 1);
    size_t got = fread(v.data(), 4, (size_t)bytes / 4, fp);
    fclose(fp);
    v.resize(got);
    return true;
}

int main(int argc, char **argv) {
    if (argc != 3) return 2;
    std::vector<int32_t> R, S;
    if (!slurp(argv[1], R) || !slurp(argv[2], S)) return 3;
    R.reserve(R.size() + 1);
    S.reserve(S.size() + 1);
    joinCpu(R.data(), R.size(), S.data(), S.size());
    return 0;
}
