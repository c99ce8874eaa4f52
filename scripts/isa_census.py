#!/usr/bin/env python3
"""Instruction census of the device ISA (`hipcc --cuda-device-only -S`): per function, and per loop of a function, how many VALU /
SALU / LDS / VMEM / SMEM / branch / wait instructions the code holds (STATIC counts; the dynamic ones are the SQ_INSTS_* counters
of profiles/r06_pmc_*).  Loops are the regions the assembler comments mark ("Loop Header: Depth=n" ... the label the back edge jumps
to); the visit's trip loop is the Depth=2 loop of wg_query_fn.

    hipcc --offload-arch=gfx950 -O3 ... --cuda-device-only -S -o /tmp/isa/nirrt_hip.s nirrt_star_amd/csrc/nirrt_hip.hip
    python scripts/isa_census.py /tmp/isa/nirrt_hip.s [name substring ...]
"""
import re
import subprocess
import sys

MIN_LOOP = 8
CLASSES = ["valu", "valu_f64", "valu_trans", "lane", "salu", "lds", "vmem", "smem", "branch", "wait", "call"]


def classify(op):
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane", "v_mov_b32_dpp")) or "_dpp" in op:
        return "lane"
    if op.startswith("v_"):
        if op.startswith(("v_rsq", "v_rcp", "v_sqrt", "v_div_", "v_exp", "v_log", "v_sin", "v_cos")):
            return "valu_trans"
        if "f64" in op or op.startswith(("v_lshl_add_u64", "v_mov_b64")):
            return "valu_f64"
        return "valu"
    if op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_barrier"):
        return "wait"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith(("s_swappc", "s_setpc")):
        return "call"
    if op.startswith(("s_load", "s_buffer_load", "s_store")):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "flat_", "scratch_", "buffer_")):
        return "vmem"
    return "salu"


def census(lines):
    c = dict.fromkeys(CLASSES, 0)
    for s in lines:
        c[classify(s.split()[0])] += 1
    return c


def main():
    path, pats = sys.argv[1], sys.argv[2:]
    funcs, cur, body = {}, None, None
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur, body = m.group(1), []
            funcs[cur] = body
            continue
        if cur is None:
            continue
        s = line.strip()
        if s.startswith(".Lfunc_end"):
            cur = None
            continue
        if not s:
            continue
        if s.startswith(";"):
            # the loop comments of a label continue on the lines after it ("Parent Loop ...", "=> This Inner Loop Header: Depth=3")
            if body and body[-1][0] == "label":
                body[-1] = ("label", body[-1][1], body[-1][2] + line)
            continue
        if s.startswith(".L") and ":" in s:
            body.append(("label", s.split(":")[0], line))
            continue
        if s.startswith("."):
            continue
        body.append(("ins", s, line))
    names = list(funcs)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    hdr = "%-74s %6s " % ("function / loop", "total") + " ".join("%8s" % c for c in CLASSES)
    print(hdr)
    for n, d in zip(names, dem):
        if pats and not any(p in d for p in pats):
            continue
        body = funcs[n]
        ins = [b[1] for b in body if b[0] == "ins"]
        c = census(ins)
        print("%-74s %6d " % (d[:74], len(ins)) + " ".join("%8d" % c[k] for k in CLASSES))
        # loops: a label whose comment says "Loop Header: Depth=k"; its extent = up to the last branch back to it
        labels = {b[1]: i for i, b in enumerate(body) if b[0] == "label"}
        for lab, i0 in labels.items():
            raw = body[i0][2]
            m = re.search(r"This (?:Inner )?Loop Header: Depth=(\d+)", raw)
            if not m:
                continue
            last = None
            for j in range(i0, len(body)):
                if body[j][0] == "ins" and body[j][1].split()[0].startswith(("s_cbranch", "s_branch")) and body[j][1].split()[-1] == lab:
                    last = j
            if last is None:
                continue
            li = [b[1] for b in body[i0:last + 1] if b[0] == "ins"]
            if len(li) < MIN_LOOP:
                continue
            lc = census(li)
            print("%-74s %6d " % ("    loop %s (depth %s)" % (lab, m.group(1)), len(li)) + " ".join("%8d" % lc[k] for k in CLASSES))


if __name__ == "__main__":
    main()
