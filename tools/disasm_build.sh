#!/bin/bash
# ISA of a built library, no GPU needed:   tools/disasm_build.sh <library.so> [kernel-name pattern]      (e.g. 'k_trace_raysILb1ELb1ELb1E')
# Without a pattern: the kernels in the library's gfx950 code object with their register / scratch / LDS use.  With one: the disassembly of the matching kernels.
set -e
LIB=$(readlink -f "${1:?library}"); PAT=$2
L=/opt/rocm/lib/llvm/bin
T=$(mktemp -d); trap 'rm -rf $T' EXIT
python3 - "$LIB" "$T/dev.co" <<'PY'
import struct, sys
b = open(sys.argv[1], "rb").read()
i = b.find(b"__CLANG_OFFLOAD_BUNDLE__")
if i < 0: sys.exit("no offload bundle in " + sys.argv[1])
n = struct.unpack_from("<Q", b, i + 24)[0]; p = i + 32
for _ in range(n):
    off, size, tl = struct.unpack_from("<QQQ", b, p); p += 24
    triple = b[p:p + tl].decode(); p += tl
    if "gfx950" in triple and size:
        open(sys.argv[2], "wb").write(b[i + off:i + off + size]); break
else:
    sys.exit("no gfx950 code object")
PY
if [ -z "$PAT" ]; then
  $L/llvm-readelf --notes $T/dev.co | grep -E "\.name:|\.vgpr_count|\.sgpr_count|private_segment_fixed_size|group_segment_fixed_size|\.vgpr_spill_count" | paste - - - - - - | sed 's/  */ /g' | sort
else
  $L/llvm-objdump -d --disassemble-symbols="$($L/llvm-objdump -t $T/dev.co | awk '{print $NF}' | grep -E "$PAT" | grep -v '\.kd$' | paste -sd,)" $T/dev.co
fi
