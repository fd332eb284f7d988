#!/usr/bin/env python3
"""Static instruction budget of the hot loop of giant_pair2_kernel<2, false, true> (the default tile kernel: quad chain; ISA_KERNEL=..ELb0EEv8TileArgs for the pair chain).

  tools/isa_budget.py [out.json]      (needs hipcc; cross-compiles for gfx950, no GPU)

Compiles csrc/tile_lines64.hip to ISA, finds the probe loop (the largest loop of the kernel), and classifies every basic block of it:
  M    a 256x256-bit multiplication with its fold (>= 60 v_mad_u64_u32)
  S    the low-64-bit squaring path (20..59 multiply-adds)
  glue everything else on the main path (field additions, probe issue / compare, addressing, loop control)
  rare blocks that only run for lanes with an exceptional event (hit reporting, equal-x doubling, exact-fold fallbacks): recognised
       by position -- the compiler places them behind the loop's back edge or guards them with an exec-mask branch -- and listed
       separately, not counted.
Per class it counts VALU instructions by cost group (sustained issue cost on MI355X, profiles/r01h_power_ops.jsonl):
  mad64   v_mad_u64_u32                                     4.2 cycles per wave instruction per SIMD
  carry   v_add_co / v_addc_co / v_sub_co / v_subb_co       4.1
  plain   every other VALU instruction                      2.3
The PMC passes (tools/profile_round.sh) give the dynamic totals; this file says what they are made of.
"""
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = os.environ.get("ISA_KERNEL", "_Z18giant_pair2_kernelILi2ELb0ELb1EEv8TileArgs")
# the translation unit that instantiates it: <2, ..> tile_lines64.hip, <3, ..> tile_lines128.hip, <4, ..> tile_lines64_any.hip (ISA_KERNEL=_Z18giant_pair2_kernelILi4ELb0ELb1EEv8TileArgs)
TU = {"2": "tile_lines64.hip", "3": "tile_lines128.hip", "4": "tile_lines64_any.hip"}[re.search(r"kernelILi(\d)", KERNEL).group(1)]
COST = {"mad64": 4.2, "carry": 4.1, "plain": 2.3}


def group(mn):
    if mn == "v_mad_u64_u32":
        return "mad64"
    if re.match(r"v_(addc?|subb?|subbrev|subrev)_co_u32", mn):
        return "carry"
    return "plain"


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else None
    asm = "/tmp/bsgs_isa_budget.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", *os.environ.get("ISA_DEFS", "").split(), "-S", "--cuda-device-only", "-o", asm,
                           os.path.join(ROOT, "bsgs-cuda_amd", "csrc", TU)], stderr=subprocess.DEVNULL)
    text = open(asm).read()
    lines = text.split("\n")
    a = next(i for i, l in enumerate(lines) if l.startswith(KERNEL + ":"))
    b = next(i for i in range(a, len(lines)) if ".end_amdhsa_kernel" in lines[i])
    body = lines[a:b]
    vgpr = re.search(re.escape(KERNEL) + r"\.num_vgpr, max\((\d+)", text)
    lab = re.compile(r"^(\.LBB\d+_\d+):")
    best = None
    for h, l in enumerate(body):
        m = lab.match(l)
        if not (m and "Loop Header" in l):
            continue
        back = [i for i, x in enumerate(body) if i > h and re.search(r"s_c?branch\w*\s+" + re.escape(m.group(1)) + r"$", x.strip())]
        if back and (best is None or back[-1] - h > best[1] - best[0]):
            best = (h, back[-1])
    start, end = best
    blocks, cur = [], None
    for i in range(start, end + 1):
        m = lab.match(body[i])
        if m:
            cur = {"label": m.group(1), "ins": []}
            blocks.append(cur)
            continue
        t = body[i].strip()
        if t and not t.startswith(";") and not t.startswith("."):
            cur["ins"].append(t.split()[0])
    # lane_moves = v_readlane_b32 / v_writelane_b32: what an SGPR the register allocator could not keep costs inside the loop (it parks scalars in lanes of a VGPR; the
    # compiler remark "SGPRs Spill" counts the parked registers, this counts the instructions that move them, per class of block)
    classes = collections.defaultdict(lambda: {"blocks": 0, "valu": 0, "mad64": 0, "carry": 0, "plain": 0, "lane_moves": 0, "s_nop": 0, "lds": 0, "vmem": 0})
    rare_tail = False
    for k, blk in enumerate(blocks):
        c = collections.Counter(blk["ins"])
        mads = c.get("v_mad_u64_u32", 0)
        valu = sum(v for n, v in c.items() if n.startswith("v_"))
        # the equal-x doubling path: a full-width squaring (49 multiply-adds) or the Px^2 squaring (45), only entered under an exec branch
        if mads >= 60:
            kind = "M"
        elif 40 <= mads < 60:
            kind = "rare"
        elif 20 <= mads < 40:
            kind = "S"
        else:
            kind = "glue"
        # hit-report blocks: global store + atomic, reached only when a ballot is non-zero
        if kind == "glue" and any(n.startswith("global_atomic") or n.startswith("global_store") for n in c):
            kind = "rare"
        # everything between the start of the equal-x path and the next block that touches memory (the prefetch / probe issue of
        # the main path) belongs to that path: its two field additions, its multiplication, its full-width squaring
        has_mem = any(n.startswith("ds_") or n.startswith("global_") or n.startswith("buffer_") for n in c)
        if rare_tail and not has_mem:
            kind = "rare"
        else:
            rare_tail = False
        if 40 <= mads < 60:
            rare_tail = True
        blk["kind"] = kind
        d = classes[kind]
        d["blocks"] += 1
        d["valu"] += valu
        d["s_nop"] += c.get("s_nop", 0)
        d["lane_moves"] += c.get("v_readlane_b32", 0) + c.get("v_writelane_b32", 0)
        d["lds"] += sum(v for n, v in c.items() if n.startswith("ds_"))
        d["vmem"] += sum(v for n, v in c.items() if n.startswith("global_") or n.startswith("buffer_"))
        for n, v in c.items():
            if n.startswith("v_"):
                d[group(n)] += v
    mode = re.search(r"kernelILi(\d)", KERNEL).group(1)
    whole = collections.Counter(x.strip().split()[0] for x in body if x.strip() and not x.strip().startswith((";", ".")))
    res = {"kernel": "giant_pair2_kernel<%s, false, %s>" % (mode, "true" if KERNEL.endswith("ELb1EEv8TileArgs") else "false"), "translation_unit": TU,
           "lane_moves_whole_kernel": {"v_readlane_b32": whole.get("v_readlane_b32", 0), "v_writelane_b32": whole.get("v_writelane_b32", 0)}, "vgprs": int(vgpr.group(1)) if vgpr else None,
           "loop": "one iteration = four giants = 8 giant steps (quad chain) or one pair of giants = 4 giant steps (pair chain); two x coordinates per giant",
           "cost_cycles_per_wave_instruction": COST, "classes": {}}
    for kind, d in classes.items():
        d["issue_cycles"] = round(sum(d[g] * COST[g] for g in COST), 1)
        res["classes"][kind] = d
    main_path = [res["classes"][k] for k in ("M", "S", "glue") if k in res["classes"]]
    steps_per_iteration = 8.0 if KERNEL.endswith("ELb1EEv8TileArgs") else 4.0      # quad chain: one iteration = four giants; pair chain: two
    per_step = {g: sum(d[g] for d in main_path) / steps_per_iteration for g in ("valu", "mad64", "carry", "plain", "lane_moves")}
    res["probe_loop_per_giant_step"] = {k: round(v, 1) for k, v in per_step.items()}
    res["probe_loop_per_giant_step"]["issue_cycles"] = round(sum(per_step[g] * COST[g] for g in COST), 1)
    m = res["classes"].get("M")
    if m:
        res["one_multiplication"] = {g: round(m[g] / m["blocks"], 1) for g in ("valu", "mad64", "carry", "plain")}
    s_ = res["classes"].get("S")
    if s_:
        res["one_low64_squaring"] = {g: round(s_[g] / s_["blocks"], 1) for g in ("valu", "mad64", "carry", "plain")}
    res["note"] = ("outside this loop every giant also costs one prefix-product multiplication + one field addition (phase 1) and 70/1024 of a "
                   "Fermat inversion (one per block of four waves): + ~0.53 multiplications per giant step; the PMC total (profiles/*_pmc_traffic.json) covers everything")
    txt = json.dumps(res, indent=1)
    if out:
        open(out, "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main()
