#!/bin/bash
# prints VGPR / SGPR / scratch / LDS of every kernel in libmjhip.so (code-object notes): tools/kernel_resources.sh [lib]
LIB=${1:-mujoco_sim_amd/libmjhip.so}
T=$(mktemp -d)
cd $T
/opt/rocm/lib/llvm/bin/clang-offload-bundler --list --type=o --input=$OLDPWD/$LIB >/dev/null 2>&1
/opt/rocm/bin/roc-obj-ls $OLDPWD/$LIB 2>/dev/null | head -0
python3 - "$OLDPWD/$LIB" <<'PY'
import sys, subprocess, re, os
lib = sys.argv[1]
data = open(lib, 'rb').read()
# find embedded ELF code objects for amdgcn: search for the offload bundle entries
idx = 0; n = 0
magic = b'\x7fELF'
outs = []
while True:
    i = data.find(magic, idx)
    if i < 0: break
    # e_machine at offset 18 (2 bytes): 224 = EM_AMDGPU
    if data[i+18:i+20] == (224).to_bytes(2, 'little'):
        # size: e_shoff + e_shnum * e_shentsize
        shoff = int.from_bytes(data[i+40:i+48], 'little'); shentsize = int.from_bytes(data[i+58:i+60], 'little'); shnum = int.from_bytes(data[i+60:i+62], 'little')
        size = shoff + shentsize * shnum
        fn = f'co{n}.elf'; open(fn, 'wb').write(data[i:i+size]); outs.append(fn); n += 1
    idx = i + 4
for fn in outs:
    txt = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-readelf', '--notes', fn], capture_output=True, text=True).stdout
    cur = {}
    for line in txt.splitlines():
        m = re.match(r'\s+- \.agpr_count:\s+(\d+)', line) or None
        for key in ('.name', '.vgpr_count', '.agpr_count', '.sgpr_count', '.private_segment_fixed_size', '.group_segment_fixed_size', '.vgpr_spill_count'):
            mm = re.match(r'\s+-?\s*' + re.escape(key) + r':\s+(\S+)', line)
            if mm:
                if key == '.name' and cur.get('.name'):
                    pass
                cur[key] = mm.group(1)
        if re.match(r'\s+\.wavefront_size', line) or re.match(r'\s+- \.args', line):
            pass
    # simpler: split per kernel on '.name:'
    blocks = re.split(r'\n\s+- \.agpr_count:', '\n' + txt)
    for b in blocks[1:]:
        b = '    - .agpr_count:' + b
        g = lambda k: (re.search(re.escape(k) + r':\s+(\S+)', b) or [None, '?'])[1]
        print(f"{g('.name')[:100]:100s} vgpr {g('.vgpr_count'):>4} agpr {g('.agpr_count'):>3} sgpr {g('.sgpr_count'):>3} scratch {g('.private_segment_fixed_size'):>5} spill {g('.vgpr_spill_count'):>3} lds {g('.group_segment_fixed_size')}")
PY
rm -rf $T
