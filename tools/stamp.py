#!/usr/bin/env python
"""Refresh feathercnn_amd/_build_stamp.json (git head + source fingerprint of the running tree; see feathercnn_amd/provenance.py).
Run where git exists -- __graft_entry__.build() does, and so does the post-commit hook `tools/stamp.py --install-hook` installs -- so the
snapshot gpurun ships to the GPU box (no .git there) knows which commit it is."""
import os
import stat
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

HOOK = """#!/bin/sh
# written by tools/stamp.py --install-hook: keep feathercnn_amd/_build_stamp.json at the new HEAD (never fails the commit)
python "$(git rev-parse --show-toplevel)/tools/stamp.py" >/dev/null 2>&1 || true
"""

if __name__ == "__main__":
    from feathercnn_amd import provenance
    if "--install-hook" in sys.argv:
        path = os.path.join(ROOT, ".git", "hooks", "post-commit")
        with open(path, "w") as f:
            f.write(HOOK)
        os.chmod(path, os.stat(path).st_mode | stat.S_IXUSR | stat.S_IXGRP | stat.S_IXOTH)
    print(provenance.write_stamp())
