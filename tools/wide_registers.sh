#!/bin/bash
# Register / spill table of every ekv_attn_wide_kernel instance (device-only ISA of each translation unit, no GPU needed):
#   tools/wide_registers.sh > profiles/<tag>_wide_kernel_registers.txt
cd "$(dirname "$0")/../easykv_amd/csrc"
echo "# hipcc --offload-arch=gfx950 -O3 -ffp-contract=off --cuda-device-only -S: .vgpr_count / .vgpr_spill_count / .sgpr_spill_count / .private_segment_fixed_size per instance"
echo "# template arguments <NWQ, NWK, KPW, REP>: <4,1,2,.> 65..128 rows, <2,2,1,.> 33..64 rows; mode 0 = one pass (output + row statistics), mode 2 = column-sum pass; REP = GQA factor (0: 8 / 16 at run time; -1: the RoPE one pass that exports logits)"
for f in ekv_attn_wide_d128_m0 ekv_attn_wide_d128_m2 ekv_attn_wide_d64_m0 ekv_attn_wide_d64_m2 ekv_attn_wide_rope_d128_m0 ekv_attn_wide_rope_d128_m2 ekv_attn_wide_rope_d64_m0 ekv_attn_wide_rope_d64_m2; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-function --cuda-device-only -S $f.hip -o /tmp/wr_$f.s 2>/dev/null
  python3 - "$f" <<'PY'
import re, sys
f = sys.argv[1]
s = open(f"/tmp/wr_{f}.s").read()
meta = s[s.index("amdhsa.kernels"):]
for blk in meta.split("  - .agpr_count")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    if "ekv_attn_wide_kernel" not in name:
        continue
    g = lambda k: re.search(rf"\.{k}:\s+(\d+)", blk).group(1)
    t = re.search(r"kernelILi(\d)ELi(\d)ELi(\d)ELi(n?\d)E", name)
    if t is None:
        print(f"{f:32s} {name}: unparsed template arguments"); continue
    print(f"{f:32s} <{','.join(x.replace('n', '-') for x in t.groups())}>  vgpr {g('vgpr_count'):>3s}  vgpr_spill {g('vgpr_spill_count')}  sgpr_spill {g('sgpr_spill_count'):>2s}  scratch_bytes {g('private_segment_fixed_size')}")
PY
done
