#!/usr/bin/env python3
"""Build-time safety check for the hand-counted s_waitcnt vmcnt(N) in conv.hip (run by `make check`).

Both position-embedding kernels wait for "everything except the N youngest vector-memory operations".  That is sound
only if, between the first LDS-DMA of the kernel and the end of the tap loop, the ONLY vector-memory operations are the
LDS-DMAs the counts were written for (plain loads issued before them are older and harmless; a plain load or a store the
compiler moved in between would shift every count).  Checked on the compiled ISA of the production instantiations:
  split-tap kernel : 3 activation DMAs + 3 weight tiles x 8 before the first barrier, then 5 x 8 inside the loop;
  ring kernel <4>  : 3 activation DMAs + 3 tiles x 2 before the first barrier, then 28 x 2 inside the loop;
and no global/flat/buffer load or store between the first DMA and the last counted wait.   usage: check_convpos.py [conv.hip]"""
import os, re, subprocess, sys, tempfile

src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "f5e-tts_amd",
                                                         "csrc", "conv.hip")
hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
with tempfile.TemporaryDirectory() as td:
    out = os.path.join(td, "c.s")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-mllvm",
                           "-amdgpu-mfma-vgpr-form", "-S", "--cuda-device-only", "-Wno-unused-value", src, "-o", out])
    lines = open(out).read().split("\n")

want = {"convpos_split_kernelILb0E": (27, 40), "convpos_kernelILi4ELb0E": (9, 56)}
cur, body = None, {}
for ln in lines:
    m = re.match(r"(_ZN12_GLOBAL__N_1\d+(convpos_\w+?)EEvNS_11ConvPosArgsE):", ln)
    if m:
        cur = m.group(2)
        body[cur] = []
        continue
    if cur is not None:
        if ln.startswith(".Lfunc_end"):
            cur = None
        else:
            body[cur].append(ln.split(";")[0].strip())

VMEM = re.compile(r"^(global|flat|buffer|scratch)_(load|store|atomic)")
ok = True
for name, (pre, loop) in want.items():
    ins = body.get(name)
    if ins is None:
        print(f"check_convpos: kernel {name} not found in the ISA")
        ok = False
        continue
    first = next(i for i, s in enumerate(ins) if s.startswith("global_load_lds"))
    waits = [i for i, s in enumerate(ins) if re.match(r"s_waitcnt vmcnt\(\d+\) lgkmcnt\(0\)", s)]
    bar = next(i for i, s in enumerate(ins) if s.startswith("s_barrier"))
    # the last counted wait of the tap loop is the last `vmcnt(0) lgkmcnt(0)` before the epilogue's first store
    store = next(i for i, s in enumerate(ins) if s.startswith("global_store"))
    last = max(i for i in waits if i < store)
    region = ins[first:last]
    dma_pre = sum(s.startswith("global_load_lds") for s in ins[first:bar])
    dma_loop = sum(s.startswith("global_load_lds") for s in ins[bar:last])
    foreign = [s for s in region if VMEM.match(s) and not s.startswith("global_load_lds")]
    good = dma_pre == pre and dma_loop == loop and not foreign
    print(f"check_convpos: {name}: {dma_pre} DMAs before the first barrier (want {pre}), {dma_loop} in the loop (want {loop}), "
          f"{len(foreign)} other vector-memory operations in between -> {'ok' if good else 'MISMATCH'}")
    ok &= good
sys.exit(0 if ok else 1)
