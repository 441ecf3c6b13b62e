"""Build-time check of the hand-scheduled prefetch in k_raster_dense (smilify_amd/csrc/raster.hip, `inflight_load`).

Pass 1 issues its next chunk's loads by inline assembly and waits for them by hand with `s_waitcnt vmcnt(n)`, because the compiler
cannot count the record stores issued in between (one in-order vmcnt on gfx950).  The compiler does not know those registers are in
flight, so the built code must be checked: between a request (`INFLIGHT_REQ`) and the statement that hands the values over
(`INFLIGHT_LANDED`), and on every way out (`INFLIGHT_DRAIN`), NO instruction may read or write the destination registers - no copy,
no spill, no reuse.  This script compiles raster.hip to gfx950 assembly and verifies exactly that for every instantiation of the
kernel; `__graft_entry__.build()` runs it and fails the build otherwise.

    python tools/check_inflight.py [--asm file.s]
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOK = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs_of(text):
    out = set()
    for m in TOK.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def check_function(name, lines):
    """lines: the function's instruction lines (comments kept).  Returns (requests, chain registers)."""
    code = [(i, ln.split(";")[0]) for i, ln in enumerate(lines)]
    req = [i for i, ln in enumerate(lines) if "INFLIGHT_REQ" in ln]
    landed = [i for i, ln in enumerate(lines) if "INFLIGHT_LANDED" in ln]
    drain = [i for i, ln in enumerate(lines) if "INFLIGHT_DRAIN" in ln]
    if not req:
        return 0, set()
    assert len(landed) == 1 and len(drain) == 1, f"{name}: expected one LANDED and one DRAIN marker, found {len(landed)} / {len(drain)}"
    landed, drain = landed[0], drain[0]
    pro = [i for i in req if i < landed]
    loop = [i for i in req if landed < i < drain]
    assert pro and loop and len(pro) == len(loop) and not [i for i in req if i > drain], f"{name}: unexpected layout of the requests {req} around {landed} / {drain}"
    dst = lambda i: regs_of(lines[i].split(";")[0].split(",")[0])  # noqa: E731  (first operand of the load)
    chain = set()
    for i in req:
        chain |= dst(i)
    assert {frozenset(dst(i)) for i in pro} == {frozenset(dst(i)) for i in loop}, f"{name}: prologue and loop requests land in different registers"
    bad = []
    for i, text in code:
        if i < pro[0] or i > drain or i in req or not text.strip() or text.strip().endswith(":"):
            continue
        used = regs_of(text) & chain
        if not used:
            continue
        if i < pro[-1]:  # between the prologue's requests: a register may be prepared before ITS request, never touched after it
            issued = set()
            for r in pro:
                if r < i:
                    issued |= dst(r)
            if used & issued:
                bad.append((i, lines[i]))
        elif landed < i < loop[0]:  # the values have landed: they are taken out here, by moves that only READ the chain
            ops = text.split(",")
            if not text.strip().startswith("v_mov_b32") or regs_of(ops[0]) & chain:
                bad.append((i, lines[i]))
        else:
            bad.append((i, lines[i]))
    # the region [first request, drain] must be closed under control flow: the rule above reads the listing top to bottom, so no block
    # that runs between a request and its hand-over may live outside the region, and nothing outside may jump into it
    labels = {ln.split(":")[0].strip(): i for i, ln in enumerate(lines) if re.match(r"^\.LBB\w+:", ln)}
    for i, text in code:
        m = re.match(r"\s*s_c?branch\w*\s+(\.LBB\w+)", text)
        if not m or m.group(1) not in labels:
            continue
        t = labels[m.group(1)]
        inside, t_inside = pro[0] < i < drain, pro[0] < t <= drain
        if inside and not (pro[0] < t <= drain + 4) and not t > drain:  # (leaving forward past the drain is impossible by construction; backward out = out of the region)
            bad.append((i, lines[i] + "   <- leaves the checked region backwards"))
        if inside and t > drain + 4:
            bad.append((i, lines[i] + "   <- jumps past the drain"))
        if not inside and t_inside and t < drain:
            bad.append((i, lines[i] + "   <- jumps into the checked region"))
    assert not bad, f"{name}: the in-flight registers {sorted(chain)} are touched between request and hand-over:\n" + "\n".join(f"  {i}: {ln.strip()}" for i, ln in bad[:20])
    return len(req), chain


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--asm", help="check this assembly file instead of compiling raster.hip")
    args = ap.parse_args()
    if args.asm:
        text = open(args.asm).read()
    else:
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        with tempfile.TemporaryDirectory() as tmp:
            out = os.path.join(tmp, "raster.s")
            subprocess.check_call([hipcc, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + os.path.join(REPO, "include"), "--cuda-device-only",
                                   "-S", os.path.join(REPO, "smilify_amd", "csrc", "raster.hip"), "-o", out], stderr=subprocess.DEVNULL)
            text = open(out).read()
    funcs, cur, name = {}, None, None
    for ln in text.splitlines():
        m = re.match(r"^(_Z\w*k_raster_dense\w*):", ln)
        if m:
            name, cur = m.group(1), []
            continue
        if cur is not None:
            if ln.strip().startswith(".Lfunc_end"):
                funcs[name], cur = cur, None
            else:
                cur.append(ln)
    assert funcs, "no k_raster_dense instantiation found in the assembly"
    total = 0
    for name, lines in sorted(funcs.items()):
        n, chain = check_function(name, lines)
        total += n
        print(f"{name}: {n} requests, chain registers {sorted(chain)}: clean")
    assert total, "no INFLIGHT_REQ marker found: the hand-scheduled prefetch is not in the build"
    return 0


if __name__ == "__main__":
    sys.exit(main())
