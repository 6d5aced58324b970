"""Build-time audit of the kernels whose accumulators / fragments are managed by inline assembly (csrc/gemm_bf16.hip).

The compiler does not know about (1) the accumulator AGPRs a0..a127 these kernels name directly and (2) the LDS reads in
flight between an inline `ds_read*` and the hand-placed `s_waitcnt lgkmcnt(0)`.  The audit compiles the file to assembly and
fails the build if, inside one of those kernels,
  * the compiler itself touches an AGPR (v_accvgpr_write with a register source, v_accvgpr_mov) - it would be using the
    accumulators as spill space - or the kernel needs scratch;
  * any instruction other than an MFMA or another LDS read names a VGPR that is the destination of an LDS read still in flight;
  * any compiler-generated instruction names the VGPR an inline-asm returning atomic (the persistent kernel's tile ticket) is
    writing, between that atomic and the next inline-asm `s_waitcnt vmcnt(0)` in layout order (the two sit in one straight trip
    of the K loop).
"""
from __future__ import annotations

import re
import subprocess

KERNELS = ("gemm_bf16_nt_big_kernel", "gemm_bf16_nt_pers_kernel", "gemm_bf16_tn_big_kernel", "gemm_bf16_tn_group_kernel")


def _regs(tok: str):
    """v5 -> {5}; v[4:7] -> {4,5,6,7}; anything else -> empty."""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def _parse(body: str):
    """-> list of (label | None, op, toks, text, in_asm)."""
    out, in_asm = [], False
    for line in body.split("\n"):
        if "#ASMSTART" in line:
            in_asm = True
            continue
        if "#ASMEND" in line:
            in_asm = False
            continue
        t = line.split(";")[0].strip()
        if not t:
            continue
        if t.endswith(":") and not t.startswith("."):
            continue
        if re.fullmatch(r"\.LBB[\w$]+:", t):
            out.append((t[:-1], None, [], t, False))
            continue
        if t.startswith("."):
            continue
        op, _, rest = t.partition(" ")
        toks = [x.strip() for x in re.split(r"[,\s]+", rest) if x.strip()]
        out.append((None, op, toks, t, in_asm))
    return out


def audit(asm_text: str):
    """Forward data-flow over the kernel's basic blocks: the set of VGPRs written by an inline-asm LDS read that has not been
    waited for (s_waitcnt lgkmcnt(0)) yet; any non-MFMA, non-asm instruction naming one of them is reported."""
    problems = []
    for chunk in asm_text.split(".globl")[1:]:
        name = chunk.split("\n", 1)[0].strip().split()[0]
        if not any(k in name for k in KERNELS) or ".Lfunc_end" not in chunk:
            continue
        ins = _parse(chunk[: chunk.index(".Lfunc_end")])
        # the stamped (STAMP = true) instances of the persistent kernel are measurement builds (tools/gemm_stamps.py): a spilled
        # time stamp there is tolerated, everything else is checked as in the production instances
        diagnostic = "gemm_bf16_nt_pers_kernel" in name and "ELb1E" in name
        # basic blocks
        starts = {0}
        label_at = {}
        for i, (lab, op, toks, _, _) in enumerate(ins):
            if lab:
                starts.add(i)
                label_at[lab] = i
            elif op.startswith("s_cbranch") or op == "s_branch" or op == "s_endpgm":
                starts.add(i + 1)
        order = sorted(x for x in starts if x < len(ins))
        end_of = {b: (order[k + 1] if k + 1 < len(order) else len(ins)) for k, b in enumerate(order)}

        # state = (ordered, loose): `ordered` = destination register sets of the in-flight inline-asm LDS reads in issue order (LDS
        # operations return in order, so `s_waitcnt lgkmcnt(N)` retires all but the youngest N - provided no scalar load or
        # compiler-issued LDS operation shares the counter at that moment); `loose` = reads whose order was lost at a merge of
        # control flow, retired only by lgkmcnt(0)
        def run(b, state, report):
            ordered, loose = list(state[0]), set(state[1])
            foreign = state[2]          # a non-asm LGKM operation was issued while asm reads were in flight
            succ = []
            last_op = None
            for i in range(b, end_of[b]):
                lab, op, toks, text, in_asm = ins[i]
                if lab:
                    continue
                last_op = op
                if op.startswith("scratch_") and report and not diagnostic:
                    problems.append(f"{name}: scratch access `{text}`")
                if not in_asm and op.startswith("v_accvgpr") and report:
                    problems.append(f"{name}: compiler-generated AGPR traffic `{text}`")
                m = re.search(r"lgkmcnt\((\d+)\)", text) if op == "s_waitcnt" else None
                if m:
                    n = int(m.group(1))
                    if n == 0:
                        ordered, loose, foreign = [], set(), False
                    elif not foreign and not loose:
                        ordered = ordered[-n:] if len(ordered) > n else ordered
                    continue
                if in_asm:
                    if op.startswith("ds_read"):
                        ordered.append(frozenset(_regs(toks[0])))
                    continue
                if (op.startswith("s_load") or op.startswith("s_buffer_load") or op.startswith("ds_")) and (ordered or loose):
                    foreign = True
                cur = set(loose)
                for r in ordered:
                    cur |= r
                used = set()
                for x in toks:
                    used |= _regs(x)
                if report and used & cur:
                    problems.append(f"{name}: `{text}` touches VGPRs {sorted(used & cur)} whose inline-asm LDS read is still in flight")
                if op == "s_branch" or op.startswith("s_cbranch"):
                    tgt = toks[-1]
                    if tgt in label_at:
                        succ.append(label_at[tgt])
            if last_op != "s_branch" and last_op != "s_endpgm" and end_of[b] < len(ins):
                succ.append(end_of[b])
            return (tuple(ordered), frozenset(loose), foreign), succ

        def merge(a, b_):
            if a is None:
                return b_
            if a[0] == b_[0]:
                return (a[0], a[1] | b_[1], a[2] or b_[2])
            loose = set(a[1]) | set(b_[1])
            for r in list(a[0]) + list(b_[0]):
                loose |= r
            return ((), frozenset(loose), a[2] or b_[2])

        # the ticket register: linear scan (layout order)
        pending = set()
        for lab, op, toks, text, in_asm in ins:
            if lab:
                continue
            if in_asm and op.startswith("global_atomic") and "sc0" in toks:
                pending = set(_regs(toks[0]))
                continue
            if in_asm and op == "s_waitcnt" and "vmcnt(0)" in text:
                pending = set()
                continue
            if pending and not in_asm:
                used = set()
                for x in toks:
                    used |= _regs(x)
                if used & pending:
                    problems.append(f"{name}: `{text}` touches VGPRs {sorted(used & pending)} that an inline-asm atomic is still writing")

        state_in = {b: None for b in order}
        state_in[order[0]] = ((), frozenset(), False)
        work = [order[0]]
        while work:
            b = work.pop()
            out, succ = run(b, state_in[b], False)
            for s_ in succ:
                if s_ not in state_in:
                    continue
                merged = merge(state_in[s_], out)
                if merged != state_in[s_]:
                    state_in[s_] = merged
                    work.append(s_)
        for b in order:
            if state_in[b] is not None:
                run(b, state_in[b], True)
    return sorted(set(problems))


def check(hipcc: str, source: str, flags: list[str]):
    r = subprocess.run([hipcc, *flags, "-S", "--cuda-device-only", source, "-o", "-"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"asm audit: hipcc failed\n{r.stderr}")
    problems = audit(r.stdout)
    if problems:
        raise RuntimeError("asm audit of the inline-assembly GEMM kernels failed:\n  " + "\n  ".join(problems[:20]))
    return True
