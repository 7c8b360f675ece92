#!/bin/bash
# tools/kernel_isa.sh <file.hip|.cpp> [extra hipcc flags] -- compiles one source of noize_job_amd/csrc for gfx950 with the
# Makefile's flags, keeps the device assembly in /tmp/isa/<name>.s and prints, per kernel, VGPRs / SGPRs / spills / scratch /
# LDS / occupancy (the .amdhsa metadata) -- what DESIGN.md quotes for register budgets and spills.
set -e
src=$1; shift
name=$(basename "${src%.*}")
dir=$(cd "$(dirname "$0")/../noize_job_amd/csrc" && pwd)
mkdir -p /tmp/isa
extra=""
case "$name" in
  nz_flow_stream) extra="-mllvm -amdgpu-sched-strategy=max-ilp" ;;
  nz_live) extra="-mllvm -amdgpu-atomic-optimizer-strategy=None -mllvm -amdgpu-sched-strategy=max-ilp" ;;
esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize \
  -Wno-unused-function $extra "$@" -x hip --cuda-device-only -S "$dir/$(basename "$src")" -o /tmp/isa/$name.s
python3 - /tmp/isa/$name.s <<'PY'
import re, sys, subprocess
txt = open(sys.argv[1]).read()
names = re.findall(r'\.set (\S+)\.private_seg_size', txt)
blocks = re.findall(r'; Kernel info:\n(.*?)\n; COMPUTE_PGM_RSRC2:TGID_Z_EN', txt, re.S)
dem = subprocess.run(['c++filt'] + names, capture_output=True, text=True).stdout.split('\n')
spills = {}
for m in re.finditer(r'\.name:\s+(\S+)\n(.*?)\.wavefront_size', txt, re.S):
    a = re.search(r'\.sgpr_spill_count:\s*(\d+)', m.group(2)); v = re.search(r'\.vgpr_spill_count:\s*(\d+)', m.group(2))
    spills[m.group(1)] = (a.group(1) if a else '-', v.group(1) if v else '-')
for n, d, b in zip(names, dem, blocks):
    def g(k):
        r = re.search(r'; ' + k + r'\s*[:=]\s*(\d+)', b)
        return r.group(1) if r else '-'
    d = re.sub(r'^\(anonymous namespace\)::', '', re.sub(r'^void ', '', d))
    d = re.sub(r'\(.*', '', d)
    # VALU instructions in the kernel body (static count)
    m = re.search(r'\n' + re.escape(n) + r':(.*?)s_endpgm', txt, re.S)
    body = m.group(1) if m else ''
    valu = len(re.findall(r'\n\s+v_(?!readlane|readfirstlane)', body))
    print('%-72s vgpr %3s sgpr %3s lds %6s code %6s scratch %4s occ %2s sgpr_spill %3s vgpr_spill %3s static_valu %5d' % (
        d[:72], g('NumVgprs'), g('TotalNumSgprs'), g('LDSByteSize'), g('codeLenInByte'), g('ScratchSize'), g('Occupancy'),
        spills.get(n, ('-', '-'))[0], spills.get(n, ('-', '-'))[1], valu))
PY
