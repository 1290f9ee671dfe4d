#!/usr/bin/env python3
"""Build-time check of the hand-written matrix-instruction streams (kge_rank_screen_r.h and friends).

gfx90a+ wants two wait states between a VALU write of a VGPR and a matrix instruction that reads it as a source (LLVM's
GCNHazardRecognizer inserts them for its own matrix instructions: LegacyVALUWritesVGPRWaitStates).  The screening kernels issue their
v_mfma as inline assembly, where the compiler can neither see the instruction nor pad it -- a register-allocator copy (v_mov /
v_accvgpr_read of a parked fragment) placed right in front of such a statement makes the matrix instruction read a stale register:
wrong, timing-dependent results (round 6: profiles/r06y2_*).  This script compiles a .hip file to device assembly and fails if any
v_mfma has a non-matrix VALU write to one of its VGPR sources within the two instructions in front of it.

usage: check_mfma_hazards.py [file.hip ...]   (default: ampligraph_amd/csrc/kge_rank.hip); exit code 1 on a hit."""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def regs(tok):
    tok = tok.strip(",")
    m = re.match(r"v\[(\d+):(\d+)\]$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def scan(asm_text):
    """-> list of (kernel, wait states in between, writer, mfma)"""
    hits, kernel, window = [], "?", []   # window: (instruction text, wait states it provides)
    for raw in asm_text.split("\n"):
        l = raw.split(";")[0].strip()
        if not l or l.startswith("."):
            continue
        m = re.match(r"^(\S+):$", l)
        if m:
            if m.group(1).startswith("_Z"):
                kernel, window = m.group(1), []
            continue   # (a label between the copy and the matrix instruction does not add a wait state)
        toks = re.split(r"[ ,]+", l)
        op = toks[0]
        if op.startswith("v_mfma") or op.startswith("v_smfmac"):
            src = set()
            for t in toks[2:5]:
                src |= regs(t)
            between = 0
            for w, ws in reversed(window[-4:]):
                if between >= 2:
                    break
                wt = re.split(r"[ ,]+", w)
                if wt[0].startswith("v_") and not wt[0].startswith(("v_mfma", "v_smfmac", "v_cmp", "v_readlane", "v_readfirstlane")) and len(wt) > 1 and regs(wt[1]) & src:
                    hits.append((kernel, between, w, l))
                between += ws
        window.append((l, int(toks[1]) + 1 if op == "s_nop" and len(toks) > 1 and toks[1].isdigit() else 1))
        window = window[-8:]
    return hits


def device_asm(path):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics", "-I" + os.path.join(ROOT, "include"),
               "-S", "--cuda-device-only", path, "-o", out]
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return open(out).read()


def main(argv):
    files = argv[1:] or [os.path.join(ROOT, "ampligraph_amd", "csrc", "kge_rank.hip")]
    bad = 0
    for f in files:
        text = open(f).read() if f.endswith(".s") else device_asm(f)
        hits = scan(text)
        n = len(re.findall(r"^\s*v_mfma", text, re.M))
        print("%s: %d matrix instructions, %d with a VALU write of a source inside two wait states" % (os.path.basename(f), n, len(hits)))
        for k, d, w, m in hits[:20]:
            print("  %s: %d wait state(s) between: %s  ->  %s" % (k[:60], d, w, m))
        bad += len(hits)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
