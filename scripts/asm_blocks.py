"""Instruction counts per basic block of one kernel in a hipcc -S listing (VALU / SALU / VMEM / LDS).
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -S --cuda-device-only -o /tmp/rtgo.s rtgo_hip.hip
    python scripts/asm_blocks.py /tmp/rtgo.s k_shadow [min_instructions]"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
name = sys.argv[2]
thr = int(sys.argv[3]) if len(sys.argv) > 3 else 15
start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN\S*" + name + r"\S*:", l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
blocks, cur = [], None
for n, ln in enumerate(lines[start + 1:end], start + 2):
    t = ln.strip()
    m = re.match(r"^(\.LBB\d+_\d+):", t)
    if m:
        cur = {"label": m.group(1), "line": n, "v": 0, "s": 0, "mem": 0, "lds": 0, "note": ""}
        blocks.append(cur)
        continue
    if not t or t.startswith(";") or t.startswith("."):
        continue
    if cur is None:
        cur = {"label": "entry", "line": n, "v": 0, "s": 0, "mem": 0, "lds": 0, "note": ""}
        blocks.append(cur)
    op = t.split()[0]
    if op.startswith("v_"):
        cur["v"] += 1
    elif op.startswith("s_"):
        cur["s"] += 1
        if op.startswith("s_cbranch") or op == "s_branch":
            cur["note"] += " ->" + t.split()[-1]
    elif op.split("_")[0] in ("global", "buffer", "scratch", "flat"):
        cur["mem"] += 1
    elif op.startswith("ds_"):
        cur["lds"] += 1
for b in blocks:
    if b["v"] + b["s"] + b["mem"] + b["lds"] >= thr:
        print(f'{b["label"]:>10} line {b["line"]:6d}  valu {b["v"]:4d} salu {b["s"]:4d} vmem {b["mem"]:3d} lds {b["lds"]:3d} {b["note"]}')
print("total valu", sum(b["v"] for b in blocks), "salu", sum(b["s"] for b in blocks), "vmem", sum(b["mem"] for b in blocks), "lds", sum(b["lds"] for b in blocks))
