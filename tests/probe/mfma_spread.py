"""Post-pass over hipcc's gfx950 assembly: spread the MFMAs of a basic block over the VALU work that follows them.

Why ("k_edge: the two pipes of a lone wave"): the fused kernels run ONE wave per SIMD.  An in-order wave issues nothing while a burst of
back-to-back MFMAs drains (32 cycles per v_mfma_f32_32x32x16_f16), and hipcc places most MFMAs of a pipeline region in such bursts in front of the
region's VALU work: sched_group_barrier is honoured only where the region's own VALU instructions reach, and inline-asm VALU blocks (DPP scans,
fp16 splits) are invisible to it.  This tool moves instructions AFTER register allocation, on the assembly text, so it costs no registers:

    for every basic block, the MFMAs keep their order and the other instructions keep theirs; an MFMA is only ever moved DOWN (later), past
    instructions that touch none of its registers, so that consecutive MFMAs are `gap` issue slots apart where the block has the work to put
    between them.

Legality of moving MFMA X below instruction Y (Y originally after X): Y names no VGPR / AGPR of X's destination or sources (any overlap -- read or
write -- stops the move: RAW, WAR and WAW at once), and Y is not a block boundary (label, branch, anything that names EXEC, barriers, priority /
message / clock instructions).  MFMAs read no SGPR / VCC / SCC and are not memory instructions, so nothing else orders them; s_waitcnt may be
crossed downwards (the data an MFMA waits for is only "more" ready later).

Hazards (LLVM GCNHazardRecognizer rules for gfx940 / gfx950, `cdna_hip_programming.md` section 5.7):
  * producer -> MFMA operand (VALU write -> MFMA read: 2 wait states; loads: s_waitcnt): every producer is BEFORE X in the original order and stays
    there; the distance only grows.
  * MFMA D -> non-MFMA reader / writer of D: passes + 4 wait states on gfx950 (12 for the 8-pass 32x32x16).  X moves TOWARDS its consumers, so
    every move keeps at least HAZARD_MARGIN issue slots between X and the first instruction that names one of its registers, and `pad_hazards()`
    re-counts the wait states of the final stream with the recogniser's own rule (s_nop N = N + 1 states, every other instruction 1) and pads.
  * MFMA D -> next MFMA taking it whole as C (accumulate chain): 0 states, and the MFMAs keep their relative order.
  * everything between non-MFMA instructions (DPP, trans, readlane, ... hazards and the s_nop hipcc placed for them): their order is unchanged and
    an MFMA dropped between two of them only adds a wait state.

    python mfma_spread.py in.s out.s [--gap 6] [--report]
"""
from __future__ import annotations

import re
import sys

HAZARD_MARGIN = 14          # wait states kept between a moved MFMA and the first instruction naming one of its registers (rule: passes + 4 <= 12 / 20)

_REG1 = re.compile(r"\b([va])(\d+)\b")
_REGR = re.compile(r"\b([va])\[(\d+):(\d+)\]")
_INS = re.compile(r"^\s+([a-z][a-z0-9_]*)\b\s*(.*)$")
_BARRIER_MNEMONICS = (
    "s_branch", "s_cbranch", "s_endpgm", "s_barrier", "s_setpc", "s_swappc", "s_call", "s_trap", "s_sethalt", "s_sleep", "s_setprio", "s_sendmsg",
    "s_setreg", "s_getreg", "s_memtime", "s_memrealtime", "s_icache_inv", "s_dcache", "s_wakeup", "s_rfe", "s_getpc", "s_set_gpr_idx", "s_cbranch_g_fork",
    "v_cmpx", "s_waitcnt_depctr", "s_ttracedata", "s_code_end",
)


def _regs(text: str) -> set:
    out = set()
    for f, a, b in _REGR.findall(text):
        out.update((f, i) for i in range(int(a), int(b) + 1))
    for f, a in _REG1.findall(text):
        out.add((f, int(a)))
    return out


def mfma_passes(mn: str) -> int:
    """Passes (4 cycles each) of an MFMA by shape; unknown shapes get the longest (16)."""
    m = re.search(r"_(\d+)x(\d+)x(\d+)_?(\w*)$", mn)
    if not m:
        return 16
    M, N, K, ty = int(m.group(1)), int(m.group(2)), int(m.group(3)), m.group(4)
    if ty in ("f16", "bf16"):
        flop = M * N * K
        return max(1, flop // 2048)          # 32x32x16 -> 8, 16x16x32 -> 4, 32x32x8 -> 4 (K-half forms), 16x16x16 -> 2
    return 16


class Ins:
    __slots__ = ("line", "mn", "ops", "kind", "regs", "weight", "passes")

    def __init__(self, line: str):
        self.line = line
        body = line.split(";", 1)[0] if not line.lstrip().startswith(";") else ""
        m = _INS.match(body) if body.strip() else None
        self.passes = 0
        if m is None:
            # comment / marker line inside a block (;;#ASMSTART ...): stays in the non-MFMA stream, no registers, no weight
            self.mn, self.ops, self.kind, self.regs, self.weight = "", "", "meta", set(), 0.0
            return
        self.mn, self.ops = m.group(1), m.group(2)
        self.regs = _regs(self.ops)
        if self.mn.startswith(("v_mfma", "v_smfmac")):
            self.kind = "mfma"
            self.weight = 1.0
            self.passes = mfma_passes(self.mn)
        elif self.mn.startswith(_BARRIER_MNEMONICS) or "exec" in self.ops or "exec" in self.mn:
            self.kind, self.weight = "barrier", 1.0
        elif self.mn == "s_nop":
            self.kind = "other"
            try:
                self.weight = (int(self.ops.strip(), 0) + 1) * 0.25
            except ValueError:
                self.weight = 0.25
        elif self.mn.startswith("s_waitcnt"):
            self.kind, self.weight = "other", 0.25
        else:
            self.kind, self.weight = "other", 1.0

    def wait_states(self) -> int:
        if self.kind == "meta":
            return 0
        if self.mn == "s_nop":
            try:
                return int(self.ops.strip(), 0) + 1
            except ValueError:
                return 1
        return 1


def _is_block_break(line: str) -> bool:
    s = line.strip()
    if not s or s.startswith(";"):
        return False
    if s.endswith(":") or s.startswith("."):      # label or directive
        return True
    return False


def schedule_block(block: list, gap: float) -> list:
    """block: list of Ins of one basic block (no labels / directives).  Returns the re-ordered list."""
    mf = [i for i, x in enumerate(block) if x.kind == "mfma"]
    if len(mf) < 1:
        return block
    others = [x for x in block if x.kind != "mfma"]
    n_oth = len(others)
    # position of an MFMA = number of non-MFMA instructions in front of it
    pos_of = {}
    k = 0
    for i, x in enumerate(block):
        if x.kind == "mfma":
            pos_of[i] = k
        else:
            k += 1
    orig = [pos_of[i] for i in mf]
    mfmas = [block[i] for i in mf]
    cws = [0]                                  # wait states of the non-MFMA stream in front of index j
    for y in others:
        cws.append(cws[-1] + y.wait_states())
    # deadline: index (in `others`) of the first instruction after the MFMA that is a barrier or names one of its registers, minus the hazard margin
    dead = []
    for m, x in enumerate(mfmas):
        lim = n_oth
        for j in range(orig[m], n_oth):
            y = others[j]
            if y.kind == "barrier" or (y.regs and not y.regs.isdisjoint(x.regs)):
                lim = j
                break
        # also: a LATER mfma that reads this one's destination as A / B (never the case in these kernels) would need the long wait: do not move then
        # keep HAZARD_MARGIN wait states (meta lines count none) between the MFMA and that instruction
        q = lim
        while q > orig[m] and cws[lim] - cws[q] < HAZARD_MARGIN:
            q -= 1
        dead.append(q)
    for m in range(len(mfmas) - 2, -1, -1):      # MFMAs keep their order
        dead[m] = min(dead[m], dead[m + 1])
    # cumulative weight of the non-MFMA stream (issue slots)
    cum = [0.0]
    for y in others:
        cum.append(cum[-1] + y.weight)

    def advance(p: int, w: float) -> int:      # smallest q >= p with cum[q] - cum[p] >= w
        target = cum[p] + w
        q = p
        while q < n_oth and cum[q] < target:
            q += 1
        return q

    # latest positions that still leave `g` slots to each successor (backward), then earliest-feasible placement (forward)
    place = []
    prev = None
    for m in range(len(mfmas)):
        g = gap * mfmas[m - 1].passes / 8.0 if m > 0 else 0.0
        want = orig[m] if prev is None else max(orig[m], advance(prev, g))
        p = min(dead[m], want)
        p = max(p, orig[m])
        if prev is not None:
            p = max(p, prev)
        place.append(p)
        prev = p
    out = []
    mi = 0
    for j in range(n_oth + 1):
        while mi < len(mfmas) and place[mi] == j:
            out.append(mfmas[mi])
            mi += 1
        if j < n_oth:
            out.append(others[j])
    assert mi == len(mfmas) and len(out) == len(block)
    return out


def pad_hazards(block: list, stats: dict | None = None) -> list:
    """MFMA D -> non-MFMA access of D on the final order, counted like the recogniser (s_nop N = N + 1 wait states, every other instruction 1, an
    MFMA 1): where an MFMA that used to stand between a producer and its consumer has moved away, the missing states are put back as an s_nop in
    front of the consumer.  (hipcc's own output needs none: `python mfma_spread.py x.s /dev/null --gap 0 --report` prints `pads 0`.)"""
    last = {}      # reg -> (wait-state clock at the MFMA, required states)
    clock = 0
    out = []
    for x in block:
        if x.kind == "mfma":
            dst = _regs(x.ops.split(",")[0])
            clock += 1
            for r in dst:
                last[r] = (clock, x.passes + 4)
            out.append(x)
            continue
        if x.kind == "meta":
            out.append(x)
            continue
        need = 0
        for r in x.regs:
            h = last.pop(r, None)
            if h is not None:
                need = max(need, h[1] - (clock - h[0]))
        if need > 0:
            out.append(Ins("\ts_nop %d" % (need - 1)))
            clock += need
            if stats is not None:
                stats["pads"] = stats.get("pads", 0) + 1
        out.append(x)
        clock += x.wait_states()
    return out


def model_cycles(block: list) -> tuple:
    """Crude in-order issue model of ONE wave per SIMD: (cycles, cycles the wave waited for the matrix pipe)."""
    t = 0.0
    free = 0.0
    stall = 0.0
    for x in block:
        if x.kind == "meta":
            continue
        if x.kind == "mfma":
            if free > t:
                stall += free - t
                t = free
            free = t + 4.0 * x.passes
            t += 4.5
        else:
            t += 4.5 * x.weight
    return t, stall


def process(text: str, gap: float = 6.0, report: bool = False) -> str:
    lines = text.split("\n")
    out = []
    block = []
    stats = {"pads": 0, "blocks": 0, "mfma": 0, "moved": 0, "model_before": 0.0, "model_after": 0.0, "stall_before": 0.0, "stall_after": 0.0}

    def flush():
        nonlocal block
        if block:
            if any(x.kind == "mfma" for x in block):
                # a block ends at its first barrier instruction: schedule the pieces between barriers separately
                piece, res = [], []
                for x in block:
                    piece.append(x)
                    if x.kind == "barrier":
                        res.extend(_sched_piece(piece, gap, stats))
                        piece = []
                res.extend(_sched_piece(piece, gap, stats))
                out.extend(x.line for x in res)
            else:
                out.extend(x.line for x in block)
            block = []

    for ln in lines:
        if _is_block_break(ln):
            flush()
            out.append(ln)
            continue
        block.append(Ins(ln))
    flush()
    if report:
        sys.stderr.write(
            "mfma_spread: %(blocks)d blocks with MFMAs, %(mfma)d MFMAs, %(moved)d moved; issue model %(model_before).0f -> %(model_after).0f cycles, "
            "matrix-pipe stalls %(stall_before).0f -> %(stall_after).0f, pads %(pads)d\n" % stats)
    return "\n".join(out)


def _sched_piece(piece: list, gap: float, stats: dict) -> list:
    if not any(x.kind == "mfma" for x in piece):
        return piece
    res = schedule_block(piece, gap)
    stats["blocks"] += 1
    stats["mfma"] += sum(1 for x in piece if x.kind == "mfma")
    stats["moved"] += sum(1 for a, b in zip(piece, res) if a is not b and a.kind == "mfma")      # (before the pads are inserted: same length)
    res = pad_hazards(res, stats)
    b, sb = model_cycles(piece)
    a, sa = model_cycles(res)
    stats["model_before"] += b
    stats["model_after"] += a
    stats["stall_before"] += sb
    stats["stall_after"] += sa
    return res


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    gap = 6.0
    for i, a in enumerate(sys.argv):
        if a == "--gap":
            gap = float(sys.argv[i + 1])
            args = [x for x in args if x != sys.argv[i + 1]]
    src, dst = args[0], args[1]
    res = process(open(src).read(), gap=gap, report="--report" in sys.argv)
    with open(dst, "w") as f:
        f.write(res)
