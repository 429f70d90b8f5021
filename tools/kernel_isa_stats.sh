#!/bin/bash
# usage: kernel_isa_stats.sh <host object or shared library with a .hip_fatbin section> [name filter]
# Prints, per gfx950 kernel: instructions, VGPRs, SGPRs, scratch bytes, LDS bytes (from the code object's metadata and disassembly).
set -e
obj="$1"; filt="${2:-.}"
tmp=$(mktemp -d)
objcopy -O binary --only-section=.hip_fatbin "$obj" "$tmp/fatbin"
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input="$tmp/fatbin" --output="$tmp/code.co" --unbundle
/opt/rocm/lib/llvm/bin/llvm-objdump -d "$tmp/code.co" > "$tmp/dis.txt"
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$tmp/code.co" > "$tmp/notes.txt"
python3 - "$tmp" "$filt" <<'PY'
import re, sys, subprocess
tmp, filt = sys.argv[1], sys.argv[2]
counts, cur = {}, None
for line in open(tmp + "/dis.txt"):
    m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
    if m:
        cur = m.group(1); counts[cur] = 0; continue
    if cur and re.match(r"^\s+[a-z_0-9]+ ", line) and "//" in line:
        counts[cur] += 1
notes = open(tmp + "/notes.txt").read()
for blk in re.split(r"\n\s+- ", notes):
    m = re.search(r"\.name:\s+(\S+)", blk)
    if not m: continue
    name = m.group(1)
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if not re.search(filt, dem): continue
    g = lambda k: (re.search(r"\.%s:\s+(\d+)" % k, blk) or [None, "?"])[1]
    print(f"{counts.get(name, '?'):>7} instr  vgpr {g('vgpr_count'):>3}  sgpr {g('sgpr_count'):>3}  scratch {g('private_segment_fixed_size'):>4}  lds {g('group_segment_fixed_size'):>6}  {dem[:110]}")
PY
rm -rf "$tmp"
