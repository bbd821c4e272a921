#!/usr/bin/env python3
"""Build-time safety check for the hand-counted s_waitcnt vmcnt(N) in gemm_bf16.hip.

The K loop allows NPC extra outstanding vector-memory operations while the prologue's tiles are awaited (the epilogue
operands prefetched behind the first LDS-DMA stages).  That is only sound if every instantiation really issues at least
NPC unconditional vector loads between the prologue's last global_load_lds and the first counted wait.  This script
compiles the file to ISA and checks exactly that with a small control-flow walk: the minimum, over all paths, of
the unmasked plain vector loads between the prologue DMAs and the first barrier must be >= NPC.  usage: check_vmcnt.py [path/to/gemm_bf16.hip]"""
import os, re, subprocess, sys, tempfile

src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "f5e-tts_amd",
                                                         "csrc", "gemm_bf16.hip")
hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
with tempfile.TemporaryDirectory() as td:
    out = os.path.join(td, "g.s")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S",
                           "--cuda-device-only", "-Wno-unused-value", src, "-o", out])
    lines = open(out).read().split("\n")

kern = re.compile(r"_ZN12_GLOBAL__N_116gemm_bf16_kernelI((?:Li\d+E)+)EEvN8f5e_gemm8GemmArgsE:")
cur, body = None, {}
for ln in lines:
    m = kern.match(ln)
    if m:
        cur = tuple(int(x) for x in re.findall(r"Li(\d+)E", m.group(1)))
        body[cur] = []
        continue
    if cur is not None:
        if ln.startswith(".Lfunc_end"):
            cur = None
        else:
            body[cur].append(ln.strip())

LABEL = re.compile(r"(\.LBB\d+_\d+):")
LOAD = re.compile(r"(global|buffer|flat)_load_(dword|ushort|ubyte|short|sbyte|sshort)")


def min_loads(ins, start, stop):
    """Fewest plain vector loads on any control-flow path from instruction `start` to instruction `stop` (a barrier).
    Loads issued while the exec mask is narrowed (after *_saveexec / v_cmpx until the mask is restored) are not
    counted: with an empty mask they may not be issued at all."""
    label_at = {}
    for i, t in enumerate(ins):
        m = LABEL.match(t)
        if m:
            label_at[m.group(1)] = i
    best = {}
    work = [(start, 0, False)]
    result = None
    while work:
        i, n, masked = work.pop()
        while True:
            if i >= len(ins):
                break
            key = (i, masked)
            if LABEL.match(ins[i]) or i == start:
                if key in best and best[key] <= n:
                    break
                best[key] = n
            if i == stop:
                result = n if result is None else min(result, n)
                break
            t = ins[i]
            if LOAD.match(t) and not t.startswith("global_load_lds"):
                n += 0 if masked else 1
            elif re.match(r"s_(and|or|xor|andn2|orn2)_saveexec|v_cmpx", t):
                masked = True
            elif re.match(r"s_(or|mov)_b64 exec", t):
                masked = False
            m = re.match(r"s_cbranch_\w+ (\.LBB\d+_\d+)", t)
            if m:
                work.append((label_at[m.group(1)], n, masked))
            m = re.match(r"s_branch (\.LBB\d+_\d+)", t)
            if m:
                i = label_at[m.group(1)]
                continue
            if t.startswith("s_endpgm"):
                break
            i += 1
    return result


bad = 0
for key, ins in sorted(body.items()):
    BM, BN, EPI, NSTAGE, DBG, WGM, WGN, FUSE = key[:8]
    NLOAD = key[8] if len(key) > 8 else 0
    if NLOAD:
        # Role split: only the loader waves stage and count vmcnt, and their waits allow NO other outstanding operation, so
        # their code -- from the prologue's first LDS-DMA to their s_endpgm, a contiguous run the consumers never enter --
        # must hold nothing but LDS-DMAs on the vector-memory queue.
        # the loader's prologue = the first run of back-to-back LDS-DMAs (the prefetch-only workgroups' DMAs sit one per
        # exec-masked branch); everything reachable from there, along both arms of every branch, is loader code
        first = next((i for i, t in enumerate(ins) if t.startswith("global_load_lds")
                      and sum(x.startswith("global_load_lds") for x in ins[i + 1:i + 8]) >= 1), None)
        okr, msg = False, "no loader prologue found"
        if first is not None:
            label_at = {m.group(1): i for i, t in enumerate(ins) for m in [LABEL.match(t)] if m}
            seen, work = set(), [first]
            while work:
                i = work.pop()
                while i < len(ins) and i not in seen:
                    seen.add(i)
                    t = ins[i]
                    m = re.match(r"s_cbranch_\w+ (\.LBB\d+_\d+)", t)
                    if m:
                        work.append(label_at[m.group(1)])
                    m = re.match(r"s_branch (\.LBB\d+_\d+)", t)
                    if m:
                        i = label_at[m.group(1)]
                        continue
                    if t.startswith("s_endpgm"):
                        break
                    i += 1
            run = [ins[i] for i in sorted(seen)]
            other = [t for t in run if re.match(r"(global|buffer|flat|scratch)_(load|store|atomic)", t)
                     and not t.startswith("global_load_lds")]
            ndma = sum(t.startswith("global_load_lds") for t in run)
            nbar = sum(t.startswith("s_barrier") for t in run)
            okr = not other and nbar >= 1 and any(t.startswith("s_endpgm") for t in run)
            msg = f"loader code = {len(run)} instructions: {ndma} LDS-DMAs, {nbar} barrier(s), {len(other)} other vector-memory operations"
        bad += not okr
        print(f"{'ok  ' if okr else 'FAIL'} gemm_bf16_kernel<{','.join(map(str, key))}> (role split): {msg}")
        continue
    TM, TN = BM // WGM // 16, BN // WGN // 16
    pref = TM * TN <= 4
    npc = 2 * TN + 4 if FUSE == 1 else 3 * TM * TN if FUSE == 2 else (2 * TM * TN if (pref and EPI == 2) else 0)
    if npc == 0:
        continue
    first_barrier = next(i for i, t in enumerate(ins) if t.startswith("s_barrier"))
    last_dma = max(i for i, t in enumerate(ins[:first_barrier]) if t.startswith("global_load_lds"))
    count = min_loads(ins, last_dma + 1, first_barrier)
    ok = count is not None and count >= npc
    bad += not ok
    print(f"{'ok  ' if ok else 'FAIL'} gemm_bf16_kernel<{','.join(map(str, key))}>: fewest unmasked vector loads on any path from "
          f"the prologue's last LDS-DMA to the first barrier = {count} (NPC {npc})")
sys.exit(1 if bad else 0)
