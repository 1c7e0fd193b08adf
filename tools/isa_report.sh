#!/bin/bash
# Compile the kernels to gfx950 assembly (no GPU needed) and report registers / spills / scratch traffic per variant.
# Usage: tools/isa_report.sh [extra hipcc flags]   -> /tmp/isa/k.s
set -e
cd "$(dirname "$0")/../ray-tracer-challenge_amd"
mkdir -p /tmp/isa
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math --offload-arch=gfx950 -Wno-unused-function \
    --cuda-device-only -S -o /tmp/isa/k.s csrc/rtc_kernels.hip "$@" 2>/dev/null
python3 - <<'PY'
import re
txt = open('/tmp/isa/k.s').read()
meta = {}
for m in re.finditer(r'\.group_segment_fixed_size: (\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size: (\d+).*?\.sgpr_spill_count: (\d+).*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count: (\d+)', txt, re.S):
    meta[m.group(2)] = m.groups()
print("%-28s %5s %5s %6s %6s %6s %6s %6s" % ("variant<S,G,P,Pers,Q>", "vgpr", "spill", "sspill", "priv", "st", "ld", "lines"))
for name, g in meta.items():
    body = re.search(r'^' + re.escape(name) + r':[^\n]*\n(.*?)\.end_amdhsa_kernel', txt, re.S | re.M)
    b = body.group(1) if body else ''
    flags = re.search(r'k_renderILb(\d)ELb(\d)ELb(\d)ELb(\d)ELb(\d)E', name)
    print("%-28s %5s %5s %6s %6s %6d %6d %6d" % (','.join(flags.groups()) if flags else name[:28], g[4], g[5], g[3], g[2],
          len(re.findall(r'scratch_store', b)), len(re.findall(r'scratch_load', b)), b.count('\n')))
PY
