#!/usr/bin/env python3
"""Lint for the hand-counted load queue of k_logmel_h / k_logmel_h_clip (csrc/logmel_h.inc): compiles csrc/embed.hip to gfx950 assembly and checks

  1. that from the first queue load on NO compiler-generated instruction of the kernel names a register in v176..v255 -- those
     belong to the queue's inline asm (the kernel is compiled with amdgpu_num_vgpr(176) and every asm block clobbers them all);
  2. that every asm block touches the registers of ONE slot only (v[176+16d : 191+16d]), awaits it with `s_waitcnt vmcnt(16)`
     before the first read (the five fills excepted), reads it only before refilling it, and refills it with exactly four
     128-bit loads, one per fragment;
     (single `global_load_dword v255` blocks in front of the first fill are discarded prefetch loads and are allowed there only);
  3. that the blocks take the slots in cyclic order 0,1,2,3,4,0,... in program text (the two arms of a branch name the same slot
     and count once) -- with 2. this is what makes "the 16 youngest loads belong to the other four slots" true at every wait.

Why: the queue's loads are invisible to hipcc's waitcnt pass.  An earlier version kept the slots in C++ variables behind asm
operands; when the register allocator placed a refill's result in another register and copied it back, the copy read a
register whose load had not landed and the compiler reused that register while the load was still due to write it (garbage
in one build, a GPU memory fault in another).  Run after touching the kernel:    python tools/check_asm_queue.py [--hooks]
Exit status 1 if anything is reported.
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "radad_retrievalaugmenteddeepfakeaudiodetection_amd", "csrc")
KERNELS = ("k_logmel_h", "k_logmel_h_clip")
QUEUE_REGS = set(range(176, 256))
REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def kernel_text(asm, name):
    """[(instruction text without comment, inside an inline-asm block?)] of the kernel"""
    lines = asm.splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*?%d%sE\w*:" % (len(name), name), l))     # (mangled: <length><name>E)
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    out, in_asm = [], False
    for l in lines[start + 1:end + 1]:
        if "#ASMSTART" in l:
            in_asm = True
        elif "#ASMEND" in l:
            in_asm = False
        out.append((l.split(";")[0].strip(), in_asm))
    return out


def asm_blocks(lines):
    blocks, cur = [], None
    for n, (ins, in_asm) in enumerate(lines):
        if in_asm:
            if cur is None:
                cur = []
                blocks.append((n, cur))
            if ins:
                cur.append(ins)
        else:
            cur = None
    return blocks


def check_reserved(lines):
    """1.: compiler-generated instructions after the first queue load that name a queue register"""
    first = next((n for n, (ins, a) in enumerate(lines) if a and ins.startswith("global_load_dwordx4")), None)
    if first is None:
        return [(0, "no queue load found")]
    bad = []
    for n, (ins, in_asm) in enumerate(lines):
        if n < first or in_asm or not ins or ins.startswith(".") or ins.endswith(":"):
            continue
        if regs_of(ins) & QUEUE_REGS:
            bad.append((n, ins))
    return bad


def check_blocks(lines):
    """2. and 3.: returns (messages, slot order)"""
    msgs, order, fills, discards = [], [], 0, 0
    for n, ins_list in asm_blocks(lines):
        qregs = set()
        for ins in ins_list:
            qregs |= regs_of(ins) & QUEUE_REGS
        if not qregs:
            continue                                   # s_nop block, final wait
        if all(re.match(r"global_load_dword v255,", i) for i in ins_list):
            # a discarded load (k_logmel_h_clip touches the next chunk's samples): allowed only BEFORE the queue's first fill --
            # loads return in order, so the fill of v255 that follows lands after it
            if order:
                msgs.append(f"line {n}: a discarded load into v255 after the queue started")
            discards += 1
            continue
        slots = {(r - 176) // 16 for r in qregs}
        if len(slots) != 1:
            msgs.append(f"line {n}: block touches slots {sorted(slots)}")
            continue
        d = slots.pop()
        loads = [i for i in ins_list if i.startswith("global_load_dwordx4")]
        dsts = {tuple(sorted(regs_of(i.split(",")[0]))) for i in loads}
        want = {tuple(range(176 + 16 * d + 4 * w, 180 + 16 * d + 4 * w)) for w in range(4)}
        if len(loads) != 4 or dsts != want:
            msgs.append(f"line {n}: slot {d} refilled by {len(loads)} loads into {sorted(dsts)}")
        first_load = next((k for k, i in enumerate(ins_list) if i.startswith("global_load")), len(ins_list))
        reads = [k for k, i in enumerate(ins_list) if not i.startswith("global_load") and regs_of(i) & QUEUE_REGS]
        waits = ins_list[0].startswith("s_waitcnt vmcnt(16)")
        if reads and not waits:
            msgs.append(f"line {n}: slot {d} read without s_waitcnt vmcnt(16) in front")
        if reads and max(reads) > first_load:
            msgs.append(f"line {n}: slot {d} read after its refill was issued")
        if not reads and not waits:
            fills += 1
        if not order or order[-1] != d:
            order.append(d)
    if fills != 5:
        msgs.append(f"{fills} initial fills (expected 5)")
    if any(d != k % 5 for k, d in enumerate(order)):
        msgs.append(f"slot order {order} is not cyclic")
    return msgs, order


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hooks", action="store_true", help="check the -DRADAD_DEBUG_HOOKS (tools-only) variant as well")
    a = ap.parse_args()
    rc = 0
    for flags in ([[]] + ([["-DRADAD_DEBUG_HOOKS"]] if a.hooks else [])):
        with tempfile.TemporaryDirectory() as tmp:
            out = os.path.join(tmp, "embed.s")
            cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
                   "--cuda-device-only", "-S", os.path.join(CSRC, "embed.hip"), "-o", out] + flags
            subprocess.run(cmd, check=True, cwd=tmp, stderr=subprocess.DEVNULL)
            asm = open(out).read()
        for kernel in KERNELS:
            lines = kernel_text(asm, kernel)
            reserved = check_reserved(lines)
            msgs, order = check_blocks(lines)
            if flags:        # basic-block layout of the tools-only variant differs (its loops can be skipped): text order says nothing there
                msgs = [m for m in msgs if "not cyclic" not in m]
            tag = " ".join(flags) or "shipped flags"
            print(f"{kernel} [{tag}]: {len(lines)} lines, {len(asm_blocks(lines))} asm blocks, slot order {''.join(map(str, order))}")
            print(f"  {len(reserved)} compiler-generated instruction(s) naming v176..v255 after the first queue load")
            for n, ins in reserved[:20]:
                print(f"    line {n}: {ins}")
            print(f"  {len(msgs)} structural problem(s)")
            for m in msgs[:20]:
                print(f"    {m}")
            rc |= 1 if (reserved or msgs) else 0
    return rc


if __name__ == "__main__":
    sys.exit(main())
