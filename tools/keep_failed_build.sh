#!/bin/bash
# Keep what a disassembler needs of a build whose GPU suite FAILED (round-5 verdict, item 7: round 4's two wrong-answer builds were not kept, so the hunt
# of round 5 could only compare good builds with good builds).  Run ON THE GPU BOX by tools/gpu_round.sh / tools/ab_run.sh when pytest fails:
#   tools/keep_failed_build.sh <library.so> <pytest log> [more files...]
# -> gpurun_out/failed_builds/<source hash>_<library name>/ : the library itself (its gfx950 code object holds every instantiation: tools/disasm_build.sh
#    lists and disassembles them anywhere, no GPU needed), the log, the compiler's version, the environment's CRH_* knobs.  gpurun merges gpurun_out/ back;
#    profiles/failed_builds/ is where a kept build is moved to by hand (untracked: *.so is git-ignored).
cd "$(dirname "$0")/.."
LIB=${1:?library}; LOG=${2:?log}; shift 2
HASH=$(python3 - <<'PY'
import sys
sys.path.insert(0, ".")
import bench
print(bench.kernel_source_hash())
PY
)
D=gpurun_out/failed_builds/${HASH}_$(basename "$LIB" .so)
mkdir -p "$D"
cp "$LIB" "$D/" && cp "$LOG" "$D/" 2>/dev/null
for f in "$@"; do cp "$f" "$D/" 2>/dev/null; done
{ /opt/rocm/bin/hipcc --version 2>&1 | head -4; echo "source hash $HASH"; env | grep '^CRH_' ; sha256sum "$LIB"; } > "$D/build_info.txt"
grep -n "FAILED\|Error\|assert" "$LOG" | head -40 > "$D/failures.txt"
echo "kept failing build in $D"
