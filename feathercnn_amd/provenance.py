"""Which tree produced a measurement.

`roofline.traffic` (HBM bytes per launch from rocprofv3 PMC passes) cannot be collected inside the benchmark process, so bench.py reads it
from a digest committed under profiles/.  A digest describes the kernels of the tree it was profiled on; this module makes that checkable:

* `source_fingerprint()` -- sha256 over the files that decide what the kernels do and which launches a net makes: every source of the
  C-ABI library (feathercnn_amd/csrc) and the package's Python side (net planner bindings, model zoo, shape tables).  bench.py's JSON plumbing
  and the docs are deliberately not part of it.
* `tree_head()` -- the git commit of the running tree: `git rev-parse HEAD` where .git exists (the build container), else the stamp
  `feathercnn_amd/_build_stamp.json` that `__graft_entry__.build()` / tools/stamp.py / the post-commit hook wrote there and that travels to
  the GPU box with the snapshot (the box has no .git).  A stamp is only believed while its fingerprint equals the live one.

tools/summarize_prof.py writes both into `traffic.json["_meta"]`; `bench.attach_traffic` attaches a digest only when its fingerprint equals
the running tree's and the profiled batch is the measured one, and otherwise reports `traffic: null` with the reason.
"""
from __future__ import annotations

import glob
import hashlib
import json
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STAMP = os.path.join(ROOT, "feathercnn_amd", "_build_stamp.json")


def fingerprint_files():
    pats = ("feathercnn_amd/csrc/*.hip", "feathercnn_amd/csrc/*.h", "feathercnn_amd/csrc/Makefile", "feathercnn_amd/*.py",
            "include/feather_hip/*.h", "include/booster/*.h")
    out = []
    for p in pats:
        out.extend(glob.glob(os.path.join(ROOT, p)))
    return sorted(os.path.relpath(f, ROOT) for f in out)


def source_fingerprint() -> str:
    h = hashlib.sha256()
    for rel in fingerprint_files():
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(rel.encode() + b"\0" + hashlib.sha256(f.read()).digest())
    return h.hexdigest()[:16]


def _git(*args):
    try:
        r = subprocess.run(["git", "-C", ROOT, *args], capture_output=True, text=True, timeout=10)
        return r.stdout.strip() if r.returncode == 0 else None
    except (OSError, subprocess.SubprocessError):
        return None


def git_head():
    """(short head, dirty?) from git itself, or (None, None) where there is no repository (the GPU box)."""
    if not os.path.isdir(os.path.join(ROOT, ".git")):
        return None, None
    head = _git("rev-parse", "--short=12", "HEAD")
    if not head:
        return None, None
    dirty = bool(_git("status", "--porcelain", "--untracked-files=no", "--", *fingerprint_files()))
    return head, dirty


def write_stamp() -> dict:
    """Record (head, dirty, fingerprint) next to the library; called where git exists.  -> the stamp."""
    head, dirty = git_head()
    st = {"git_head": head, "git_dirty": dirty, "source_fingerprint": source_fingerprint()}
    if head:
        with open(STAMP, "w") as f:
            json.dump(st, f)
    return st


def tree_head() -> dict:
    """{"git_head", "git_dirty", "source_fingerprint", "from"} of the running tree."""
    fp = source_fingerprint()
    head, dirty = git_head()
    if head:
        return {"git_head": head, "git_dirty": dirty, "source_fingerprint": fp, "from": "git"}
    try:
        st = json.load(open(STAMP))
    except (OSError, ValueError):
        st = {}
    if st.get("source_fingerprint") == fp and st.get("git_head"):
        return {"git_head": st["git_head"], "git_dirty": st.get("git_dirty"), "source_fingerprint": fp, "from": "build stamp"}
    return {"git_head": None, "git_dirty": None, "source_fingerprint": fp, "from": "no git and no matching build stamp"}


if __name__ == "__main__":
    print(json.dumps(write_stamp()))
