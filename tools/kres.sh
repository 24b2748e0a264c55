#!/bin/bash
# usage: tools/kres.sh <file.hip> [extra flags] -- per-kernel VGPR / spill / scratch summary (cross-compile, no GPU needed)
cd /root/repo/poolgen_amd/csrc
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Rpass-analysis=kernel-resource-usage "$@" -c $f -o /tmp/kres.o 2>&1 | python3 -c '
import re, sys, subprocess
cur = None
for line in sys.stdin:
    m = re.search(r"remark: +(.*?) \[-Rpass", line)
    if not m: 
        if "error" in line: print(line.rstrip())
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        name = t.split(":",1)[1].strip()
        try: name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip()
        except Exception: pass
        name = re.sub(r"\(anonymous namespace\)::", "", name); name = re.sub(r"\(.*", "", name)
        cur = {"name": name}
    elif cur is not None:
        k, _, v = t.partition(":")
        cur[k.strip()] = v.strip()
        if k.strip().startswith("LDS Size"):
            print("%-55s VGPR %-4s spillV %-4s scratch %-5s occ %s" % (cur["name"][:55], cur.get("VGPRs"), cur.get("VGPRs Spill"), cur.get("ScratchSize [bytes/lane]"), cur.get("Occupancy [waves/SIMD]")))
            cur = None
'
