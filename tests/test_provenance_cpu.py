"""roofline.traffic may only come from a PMC digest that describes the running tree: feathercnn_amd/provenance.py
fingerprints the kernel + runtime sources, tools/summarize_prof.py stamps traffic.json with it, bench.attach_traffic attaches a digest only
when fingerprint, batch and fusion level match, and says why when they do not.  CPU-only: no GPU, no library calls."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_fingerprint_follows_the_kernel_sources_only(tmp_path, monkeypatch):
    from feathercnn_amd import provenance
    files = provenance.fingerprint_files()
    assert any(f.endswith("gemm_core.h") for f in files) and any(f.endswith("winograd_f63.hip") for f in files) and any(f.endswith("net.py") for f in files)
    assert not any("bench.py" in f or f.endswith(".md") or f.startswith("tests/") for f in files)
    a = provenance.source_fingerprint()
    assert a == provenance.source_fingerprint() and len(a) == 16
    # another tree (one changed kernel source) has another fingerprint
    fake = tmp_path / "repo"
    (fake / "feathercnn_amd" / "csrc").mkdir(parents=True)
    (fake / "feathercnn_amd" / "csrc" / "k.hip").write_text("kernel v1")
    monkeypatch.setattr(provenance, "ROOT", str(fake))
    f1 = provenance.source_fingerprint()
    (fake / "feathercnn_amd" / "csrc" / "k.hip").write_text("kernel v2")
    assert provenance.source_fingerprint() != f1 != a


def test_tree_head_prefers_git_and_believes_a_stamp_only_while_it_matches(tmp_path, monkeypatch):
    from feathercnn_amd import provenance
    here = provenance.tree_head()
    assert here["source_fingerprint"] == provenance.source_fingerprint()
    if os.path.isdir(os.path.join(ROOT, ".git")):
        assert here["from"] == "git" and here["git_head"]
    # no git: a matching stamp is believed, a stale one is not
    fake = tmp_path / "repo"
    (fake / "feathercnn_amd" / "csrc").mkdir(parents=True)
    (fake / "feathercnn_amd" / "csrc" / "k.hip").write_text("kernel")
    monkeypatch.setattr(provenance, "ROOT", str(fake))
    monkeypatch.setattr(provenance, "STAMP", str(fake / "feathercnn_amd" / "_build_stamp.json"))
    fp = provenance.source_fingerprint()
    assert provenance.tree_head()["git_head"] is None
    json.dump({"git_head": "abc123", "git_dirty": False, "source_fingerprint": fp}, open(provenance.STAMP, "w"))
    assert provenance.tree_head() == {"git_head": "abc123", "git_dirty": False, "source_fingerprint": fp, "from": "build stamp"}
    (fake / "feathercnn_amd" / "csrc" / "k.hip").write_text("kernel, edited after the stamp")
    assert provenance.tree_head()["git_head"] is None


def test_attach_traffic_only_for_the_running_tree(tmp_path, monkeypatch):
    from benchkit import roofs as bench
    from feathercnn_amd import provenance
    fp = provenance.source_fingerprint()
    prof = tmp_path / "profiles"
    (prof / "r09_vgg16").mkdir(parents=True)
    (prof / "r10_vgg16").mkdir(parents=True)
    row = {"launches_profiled": 4, "hbm_bytes_per_launch": 1000.0}
    stale = {"wino_gemm_glds_kernel<2>": dict(row, hbm_bytes_per_launch=7.0), "_meta": {"source_fingerprint": fp, "git_head": "old", "fusion": 3, "nets": {"vgg16": {"per_gpu_batch": 32}}}}
    good = {"wino_gemm_glds_kernel<2>": row, "gemm_mfma_kernel<Shape<64,128>, WinoGemmPolicy>": dict(row, launches_profiled=12, hbm_bytes_per_launch=2000.0),
            "_meta": {"source_fingerprint": fp, "git_head": "prof", "fusion": 3, "nets": {"vgg16": {"per_gpu_batch": 32}}}}
    json.dump(stale, open(prof / "r09_vgg16" / "traffic.json", "w"))
    json.dump(good, open(prof / "r10_vgg16" / "traffic.json", "w"))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    mk = lambda: [{"kernel": "Winograd tile GEMM: ...", "traffic": None}, {"kernel": "something without counters", "traffic": None}]  # noqa: E731
    r = mk()
    bench.attach_traffic("vgg16", r, batch=32, fusion=3)       # round 10 is newer than round 9 (numeric order, not "r1" < "r9")
    assert r[0]["traffic"] == round((4 * 1000.0 + 12 * 2000.0) / 16) and r[0]["traffic_profiled_at"] == "prof" and r[0]["traffic_fingerprint"] == fp
    assert r[0]["traffic_head"] == provenance.tree_head()["git_head"] and "r10_vgg16" in r[0]["traffic_source"] and r[1]["traffic"] is None
    r = mk()
    bench.attach_traffic("vgg16", r, batch=16, fusion=3)       # another batch than the profiled one
    assert r[0]["traffic"] is None and "batch" in r[0]["traffic_stale"]
    r = mk()
    bench.attach_traffic("vgg16", r, batch=32, fusion=2)       # another fusion level
    assert r[0]["traffic"] is None and "fusion" in r[0]["traffic_stale"]
    good["_meta"]["source_fingerprint"] = "0" * 16              # profiled on other sources
    json.dump(good, open(prof / "r10_vgg16" / "traffic.json", "w"))
    r = mk()
    bench.attach_traffic("vgg16", r, batch=32, fusion=3)
    assert r[0]["traffic"] is None and "profiled on sources" in r[0]["traffic_stale"]
    del good["_meta"]                                           # a digest from before round 4
    json.dump(good, open(prof / "r10_vgg16" / "traffic.json", "w"))
    r = mk()
    bench.attach_traffic("vgg16", r, batch=32, fusion=3)
    assert r[0]["traffic"] is None and "no source fingerprint" in r[0]["traffic_stale"]
