#!/usr/bin/env python3
"""
Calibration of bench.py's `cpu_baseline`: the oracle port (oracle/np_oracle.py, what the GPU
box times) against the py3 translation of the reference itself (tests/golden/build_ref.py -> /tmp/segk_ref, only
possible in the build container where /root/reference is mounted), same inputs, same seeds, one core.

    python tests/golden/build_ref.py && python tools/cpu_calibration.py [n_utts]

Prints ms/utterance of both and the ratio port / reference; the ratio is committed in profiles/cpu_calibration.json
and quoted by bench.py as cpu_baseline.calibration.
"""
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    n_utts = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    from segmentalist_amd.synth import make_corpus
    from oracle import np_oracle as no
    sys.path.insert(0, "/tmp/segk_ref")
    from segmentalist import kmeans_acoustic_wordseg as ref_kaw
    corpus = make_corpus(n_utts, 100, 1000, seed=0, N=20, n_slices_max=6)
    out = {}
    state = {}
    for name, mod in (("reference_py3_translation", ref_kaw), ("oracle_port", no)):
        random.seed(0)
        np.random.seed(0)
        seg = mod.SegmentalKMeansWordseg(1000, *corpus, n_slices_max=6, init_am_assignments="spread")
        order = list(range(seg.utterances.D))
        random.shuffle(order)
        t0 = time.perf_counter()
        for i in order:
            seg.segment_i(i)
        dt = time.perf_counter() - t0
        out[name + "_ms_per_utt"] = 1e3 * dt / n_utts
        state[name] = (seg.utterances.boundaries.copy(), seg.acoustic_model.components.assignments.copy())
    assert np.array_equal(state["reference_py3_translation"][0], state["oracle_port"][0])
    assert np.array_equal(state["reference_py3_translation"][1], state["oracle_port"][1])
    out["ratio_port_over_reference"] = out["oracle_port_ms_per_utt"] / out["reference_py3_translation_ms_per_utt"]
    out["n_utts"] = n_utts
    out["host"] = "build container, %d cores visible, 1 used" % os.cpu_count()
    out["identical_state"] = True
    print(json.dumps(out, indent=1))
    json.dump(out, open(os.path.join(ROOT, "profiles", "cpu_calibration.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
