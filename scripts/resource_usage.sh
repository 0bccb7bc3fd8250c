#!/bin/bash
# Per-kernel register / spill / LDS summary of one kernel file (cross-compiles for gfx950; no GPU needed).
# usage: scripts/resource_usage.sh chol_trsm.hip [grep-pattern]      (XFLAGS: extra compiler flags)
cd "$(dirname "$0")/../xivo_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wno-unused-result ${XFLAGS:-} \
  -Rpass-analysis=kernel-resource-usage -c "$1" -o /tmp/ru_$$.o > /tmp/ru_$$.txt 2>&1
python3 - /tmp/ru_$$.txt "${2:-.}" <<'PY'
import re, subprocess, sys
rows, cur = [], None
for line in open(sys.argv[1]):
    m = re.search(r"remark: (?:\s*)([A-Za-z \[\]/]+): (\S+)", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2)
    if k == "Function Name":
        cur = {"name": v}; rows.append(cur)
    elif cur is not None:
        cur[k] = v
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.split("\n")
for r, n in zip(rows, names):
    n = n.replace("xivo_hip::(anonymous namespace)::", "").replace("void ", "")
    n = re.sub(r"\(.*\)$", "", n)
    if re.search(sys.argv[2], n):
        print(f"{n:52s} VGPR {r.get('VGPRs'):>4} AGPR {r.get('AGPRs'):>4} spillV {r.get('VGPRs Spill'):>4} spillS {r.get('SGPRs Spill'):>4} "
              f"scratch {r.get('ScratchSize [bytes/lane]'):>5} occ {r.get('Occupancy [waves/SIMD]'):>2} lds {r.get('LDS Size [bytes/block]')}")
PY
rm -f /tmp/ru_$$.o /tmp/ru_$$.txt
