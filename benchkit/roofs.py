"""Roofline rows: algorithmic work / measured time against the gfx950 peaks; the pure-MFMA calibration; PMC traffic digests."""
from __future__ import annotations

import json
import os

from . import PEAK_HBM_GBS, PEAK_MFMA_F32_TFLOPS, ROOT

_SUSTAINED = {}


def sustained_mfma():
    """fhip_calibrate_mfma_f32, once per process: what a kernel made of nothing but fp32 MFMAs reaches on THIS device (the chip clocks
    to its power budget under full-chip matrix load), and the shader clock it ran at."""
    if not _SUSTAINED:
        from feathercnn_amd import booster
        try:
            tf, mhz = booster.calibrate_mfma_f32()
            _SUSTAINED.update({"tflops": round(tf, 1), "shader_mhz": round(mhz)})
        except Exception as e:  # a measurement aid must never take the benchmark down
            _SUSTAINED.update({"tflops": None, "error": repr(e)})
    return _SUSTAINED


def roofline_mfma(kernel, flops, ms, note):
    ach = flops / ms / 1e9
    r = {"kernel": kernel, "bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_MFMA_F32_TFLOPS, "unit": "TFLOP/s",
         "frac": round(ach / PEAK_MFMA_F32_TFLOPS, 4), "traffic": None, "work_per_step": flops, "ms_per_step": round(ms, 4), "note": note}
    sus = sustained_mfma()
    if sus.get("tflops"):
        # next to the nominal peak (2.4 GHz): the measured ceiling of a pure-MFMA kernel on this device in this process
        r["sustained_peak_measured"] = sus["tflops"]
        r["shader_mhz_under_mfma_load"] = sus["shader_mhz"]
        r["frac_of_sustained"] = round(ach / sus["tflops"], 4)
    return r


def roofline_hbm(kernel, nbytes, ms, note):
    ach = nbytes / ms / 1e6
    return {"kernel": kernel, "bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": round(ach / PEAK_HBM_GBS, 4), "traffic": None, "work_per_step": nbytes, "ms_per_step": round(ms, 4), "note": note}


# kernels behind each roofline row, as rocprofv3 names them (profiles/<round>_<net>/traffic.json keys)
TRAFFIC_KERNELS = {"Winograd tile GEMM": ("wino_gemm_glds", "WinoGemmPolicy"), "1x1 implicit GEMM": ("ConvGemmPolicy<1", "ConvGemmPolicy<2", "ConvGemmPolicy<5", "stream_gemm_kernel"),
                   "depthwise": ("depthwise3x3_",), "fused depthwise 3x3 + 1x1": ("ConvGemmPolicy<3", "ConvGemmPolicy<4", "dwpw_band_kernel"),
                   "wino_input_": ("wino_input_staged_kernel", "wino_input_transform_kernel", "wino43_input_transform_kernel", "wino_input_from_first"), "wino_chain_kernel": ("wino_chain_kernel",)}


def _round_of(path):
    """profiles/r12_vgg16/traffic.json -> 12 (numeric, so r10 sorts after r9)."""
    import re
    m = re.match(r"r(\d+)_", os.path.basename(os.path.dirname(path)))
    return int(m.group(1)) if m else -1


def attach_traffic(net_name, roofs, batch=None, sub_batches=1, fusion=None):
    """roofline.traffic: HBM bytes per launch of the row's kernels from the rocprofv3 PMC passes of THIS command (2 * FETCH_SIZE + WRITE_SIZE,
    separate --pmc passes, the gfx950 correction of MI355X_MICROARCH.md; tools/profile.sh + tools/summarize_prof.py).  PMC counters cannot be
    read inside the benchmark process, so the figure comes from the digest committed under profiles/ (newest round that has one for this net)
    -- and ONLY when that digest describes the tree that is running: its `_meta.source_fingerprint` (sha256 over the kernel and runtime
    sources, feathercnn_amd/provenance.py) must equal the live one and its profiled batch / fusion level the measured ones.  Otherwise
    traffic stays null and `traffic_stale` says why.  `traffic_head` = git commit of the running tree (the digest is valid for it because
    the fingerprints are equal), `traffic_profiled_at` = the commit the profile was taken on.  Launch-weighted mean over the kernels of
    the row; `achieved` and `frac` stay live measurements."""
    import glob
    from feathercnn_amd import provenance
    # newest round first (numerically); within a round the single-stream profile (what the per-kernel attribution runs) before the replica one
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{net_name}", "traffic.json")) +
                   glob.glob(os.path.join(ROOT, "profiles", f"r*_{net_name}_single_stream", "traffic.json")),
                   key=lambda q: (_round_of(q), q.endswith("_single_stream/traffic.json")))
    if not cands:
        return
    path = cands[-1]
    try:
        dig = json.load(open(path))
    except (OSError, ValueError):
        return
    meta = dig.get("_meta") or {}
    here = provenance.tree_head()
    stale = None
    if not meta.get("source_fingerprint"):
        stale = "the digest carries no source fingerprint (profiled before round 4)"
    elif meta["source_fingerprint"] != here["source_fingerprint"]:
        stale = f"profiled on sources {meta['source_fingerprint']} (commit {meta.get('git_head')}), running {here['source_fingerprint']}"
    else:
        prof = (meta.get("nets") or {}).get(net_name) or {}
        if batch is not None and prof.get("per_gpu_batch") not in (None, batch):
            stale = f"profiled at batch {prof.get('per_gpu_batch')}, measured at {batch}"
        elif fusion is not None and meta.get("fusion") not in (None, fusion):
            stale = f"profiled at fusion level {meta.get('fusion')}, measured at {fusion}"
    src = os.path.relpath(path, ROOT)
    for r in roofs:
        pats = next((v for k, v in TRAFFIC_KERNELS.items() if r["kernel"].startswith(k)), None)
        if not pats:
            continue
        if stale:
            r["traffic"] = None
            r["traffic_stale"] = f"{src}: {stale}"
            continue
        rows = [v for k, v in dig.items() if k != "_meta" and any(q in k for q in pats)]
        n = sum(v["launches_profiled"] for v in rows)
        if n:
            r["traffic"] = round(sum(v["hbm_bytes_per_launch"] * v["launches_profiled"] for v in rows) / n)
            r["traffic_unit"] = "HBM bytes per launch (launch-weighted mean over the row's kernels)"
            r["traffic_source"] = src + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command; 2*FETCH + WRITE)"
            r["traffic_head"] = here["git_head"]
            r["traffic_profiled_at"] = meta.get("git_head")
            r["traffic_fingerprint"] = here["source_fingerprint"]


TRAFFIC_NOTE = ("traffic (HBM bytes per launch from rocprofv3 PMC passes) cannot be collected inside this process: it is read from the digest of "
                "the same command committed under profiles/ (traffic_source) and attached only when the digest's source fingerprint "
                "(feathercnn_amd/provenance.py: sha256 over the kernel + runtime sources) equals the running tree's and the profiled batch / "
                "fusion level are the measured ones -- otherwise traffic is null and traffic_stale says why; traffic_head = commit of the running "
                "tree, traffic_profiled_at = commit the counters were collected on")
