#!/bin/bash
# Scratch (spill) instructions INSIDE the MFMA region of every eval_kernel build of a translation unit - the metric that caught a
# 30 % regression of config 5's kernel in round 3.  Device-only compile, no GPU needed:
#   bash tools/check_hot_loop_spills.sh npbnn_eval_inst_d3_cat_fast [extra compiler flags]
# prints, per kernel: VGPRs, scratch bytes, spilled VGPRs, then "mfma-region scratch ops: N of total M".
TU=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${TMPDIR:-/tmp}
LLVM=/opt/rocm/lib/llvm/bin
cd $ROOT/npbnn_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -I$ROOT/include \
  -mllvm -amdgpu-kernarg-preload-count=4 "$@" -c -o $OUT/spill_$TU.o $TU.hip || exit 1
cd $OUT && $LLVM/clang-offload-bundler --unbundle --type=o --input=spill_$TU.o --targets=hip-amdgcn-amd-amdhsa--gfx950 --output=spill_$TU.elf
$LLVM/llvm-objdump -d spill_$TU.elf --no-show-raw-insn > spill_$TU.s
$LLVM/llvm-readelf --notes spill_$TU.elf | grep -E "\.name:|\.vgpr_count|private_segment_fixed|\.vgpr_spill" | paste - - - - \
  | sed 's/_ZN5npbnn11eval_kernel//' | awk '{print $2, "scratch bytes", $4, "vgprs", $6, "spilled", $8}'
python3 - $OUT/spill_$TU.s <<'PY'
import re, sys
name, body = None, {}
for line in open(sys.argv[1]):
    m = re.match(r'^[0-9a-f]+ <(\S+)>:$', line)
    if m:
        name = m.group(1); body[name] = []; continue
    if name:
        body[name].append(line)
for k, b in body.items():
    idx = [i for i, l in enumerate(b) if 'v_mfma' in l]
    if not idx:
        continue
    n = sum(1 for l in b[idx[0]:idx[-1] + 1] if 'scratch_' in l)
    print(k.replace('_ZN5npbnn11eval_kernel', ''), 'mfma-region scratch ops:', n, 'of total', sum(1 for l in b if 'scratch_' in l))
PY
