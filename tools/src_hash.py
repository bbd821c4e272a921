#!/usr/bin/env python3
"""sha256 over the kernel sources (f5e-tts_amd/csrc/*.hip, *.h, Makefile + include/*.h), file names included, in sorted
order.  The GPU box has no .git, so profile summaries that bench.py reads back (profiles/pmc_traffic_*.json) are stamped
with this hash instead of a commit id: bench.py recomputes it and refuses a summary made from other kernel code."""
import glob
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_sha256(root: str = ROOT) -> str:
    files = sorted(glob.glob(os.path.join(root, "f5e-tts_amd", "csrc", "*.hip"))
                   + glob.glob(os.path.join(root, "f5e-tts_amd", "csrc", "*.h"))
                   + [os.path.join(root, "f5e-tts_amd", "csrc", "Makefile")]
                   + glob.glob(os.path.join(root, "include", "*.h")))
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.relpath(f, root).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


if __name__ == "__main__":
    print(csrc_sha256(sys.argv[1] if len(sys.argv) > 1 else ROOT))
