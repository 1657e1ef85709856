#!/bin/bash
# tools/gpu_kernel_coverage.sh — which device kernel instances of libhj.so does the GPU test suite launch?  Runs the suite under
# rocprofv3 --kernel-trace --stats (no counters, no other trace domain), collects the kernel names of every process that wrote a
# stats file, and prints the instances of the binary (profiles/<round>_libhj_kernels.txt) that were never launched.
cd /tmp && export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/coverage
rm -rf $OUT && mkdir -p $OUT
cd $ROOT
# Several host threads submitting to ONE hardware queue under the kernel trace's queue interception kill the process (the in-process
# multi-rank tests; profiles/r5_rocprof_suite_crash.txt): give every stream a queue of its own while profiling.
export GPU_MAX_HW_QUEUES=16
# (the profiled suite dies of a fault in about one run of six — never unprofiled: profiles/r5_rocprof_suite_crash.txt — so: up to three tries)
for try in 1 2 3; do
  rm -rf $OUT/trace
  timeout 3000 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1
  rc=$?; echo "pytest rc=$rc (try $try)"; tail -2 $OUT/pytest.log
  [ $rc -lt 128 ] && break
done
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
launched = {}
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        name = row.get("Name") or row.get("KernelName") or ""
        launched[name] = launched.get(name, 0) + int(row.get("Calls", 0) or 0)
# the instances of the binary: profiles/<round>_libhj_kernels.txt (tools/kernel_sizes.sh, written in the build container from the same sources)
dem = {}
for line in open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "profiles", os.environ.get("ROUND", "r6") + "_libhj_kernels.txt")):
    p = line.split(None, 2)
    if len(p) == 3 and p[0].isdigit():
        dem[p[2].strip()] = p[2].strip()
def base(n):   # rocprof prints demangled names without the argument list for templates in some versions: compare up to '('
    return n.split("(")[0].replace("void ", "").strip()
lb = {}
for k, v in launched.items():
    lb[base(k)] = lb.get(base(k), 0) + v
with open(out + "/kernel_coverage.txt", "w") as f:
    f.write("# kernel instances of libhj.so and how often the GPU test suite (pytest -m gpu, the pytest process and every child that wrote a\n"
            "# rocprofv3 stats file) launched them.  0 = never launched by a test.\n")
    never = []
    for m in sorted(dem, key=lambda m: dem[m]):
        c = lb.get(base(dem[m]), 0)
        f.write("%10d  %s\n" % (c, dem[m][:170]))
        if c == 0:
            never.append(dem[m])
    f.write("\n%d instances, %d never launched\n" % (len(dem), len(never)))
    other = sorted(k for k in lb if k.startswith("hj::") and k not in {base(v) for v in dem.values()})
    if other:
        f.write("\nlaunched but not matched to a symbol (name formatting):\n" + "\n".join("  " + o for o in other) + "\n")
print(open(out + "/kernel_coverage.txt").read()[-3000:])
PY
rm -rf $OUT/trace
