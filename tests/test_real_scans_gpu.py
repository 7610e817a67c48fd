"""The reference's own scans (test_data/binarize, held as data in tests/golden/scans/*.npz) through the HIP path.

Whole scans of >= 700 x 1200 pixels, 1024 x 1536 crops of the four largest scans and one 5312-column band: wide enough for
interior strips, so the float32 interior pipeline, the tiers, the refine queue and Wolf-Jolion's candidate list all see real
paper (the six small fixtures of tests/golden/ only ever reach edge strips).  Every fixture runs the five header-default calls
and the headline parameters in PRL_MODE_AUTO (fused kernel) and PRL_MODE_LITERAL; both must equal the committed oracle mask
bit for bit.  The queue statistics per image and configuration are printed and, when gpurun_out/ exists, written to
gpurun_out/real_scans.jsonl (copied to profiles/r04/real_scans.jsonl).
"""
import glob
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCANS = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "scans", "*.npz")))


def _configs(z):
    for key in z.files:
        if key.startswith("mask_"):
            name = key[5:]
            m, w, k, mo = z["params_" + name]
            yield name, int(m), int(w), float(k), int(mo), tuple(int(v) for v in z["shape_" + name])


def _want(z, name, shape):
    return np.unpackbits(z["mask_" + name], axis=1)[:, :shape[1]].astype(np.uint8) * 255


def test_fixture_set_is_the_one_the_verdict_asked_for():
    whole = [p for p in SCANS if "_x" not in os.path.basename(p)]
    assert len(SCANS) >= 24 and len(whole) >= 16
    for p in whole:
        h, w = np.load(p)["gray"].shape
        assert min(h, w) >= 700 and max(h, w) >= 1195, (p, h, w)
    assert sum(os.path.getsize(p) for p in SCANS) < 40 * 2**20


@pytest.mark.parametrize("path", SCANS, ids=[os.path.basename(p)[:-4] for p in SCANS])
def test_oracle_reproduces_the_scan_fixtures(oracle, path):
    """CPU: the committed masks are what the oracle computes today (the numpy model agreed at generation time)."""
    z = np.load(path)
    gray = z["gray"]
    for name, m, w, k, mo, shape in _configs(z):
        got = oracle.binarize(gray, oracle.make_params(m, w, k, mo))
        assert got.shape == shape and np.array_equal(got, _want(z, name, shape)), (os.path.basename(path), name)


@pytest.mark.gpu
def test_real_scans_fused_and_literal_equal_the_oracle(prl, cuda_device):
    import torch

    records = []
    for path in SCANS:
        z = np.load(path)
        gray = torch.from_numpy(z["gray"]).to(cuda_device)
        for name, m, w, k, mo, shape in _configs(z):
            want = _want(z, name, shape)
            p = prl.make_params(m, w, k, mo)
            rec = {"image": os.path.basename(path)[:-4], "height": int(gray.shape[0]), "width": int(gray.shape[1]), "config": name,
                   "window": w, "k": k, "morph": mo}
            for mode, tag in ((0, "fused"), (1, "literal")):
                prl.set_exec_mode(mode)
                try:
                    got = prl.binarize(gray, p).cpu().numpy()
                    st = prl.last_stats()
                finally:
                    prl.set_exec_mode(0)
                assert got.shape == shape, (rec, tag)
                bad = int((got != want).sum())
                assert bad == 0, f"{rec['image']} {name} {tag}: {bad} pixels differ from the oracle mask"
                if mode == 0:
                    rec.update(pixels=int(st.pixels), refined_pixels=int(st.refined_pixels), exact_pixels=int(st.exact_pixels),
                               literal_pages=int(st.literal_pages), wolf_candidates=int(st.wolf_candidates))
                    # a real scan must not fall off the fast path (flat scanner borders are the one known way to: report it)
                    rec["fell_to_literal"] = bool(st.literal_pages)
            records.append(rec)
    n_lit = sum(r["literal_pages"] for r in records)
    print(f"\nreal scans: {len(SCANS)} images x 6 configurations, fused == literal == oracle on every pixel; "
          f"{sum(r['refined_pixels'] for r in records)} refined, {sum(r['exact_pixels'] for r in records)} exact, "
          f"{n_lit} calls redone by the literal pipeline")
    for r in records:
        print(f"  {r['image']:28s} {r['config']:19s} refined {r['refined_pixels']:7d} exact {r['exact_pixels']:5d} "
              f"literal {r['literal_pages']} wolf-candidates {r['wolf_candidates']}")
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "real_scans.jsonl"), "w") as f:
            for r in records:
                f.write(json.dumps(r) + "\n")
    assert len(records) == 6 * len(SCANS)
