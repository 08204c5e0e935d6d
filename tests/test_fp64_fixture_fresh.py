"""CPU check that tests/golden/fp64_envelope.json still belongs to the inputs of the GPU envelope tests: every entry's fingerprint
(sha256 of the initial weights of the oracle model, the cohort tensors, the hyper-parameters and the text of oracle/*.py, + the torch version)
is recomputed here.  A stale entry would not make a GPU test
wrong -- it falls back to computing the fp64 leg live -- only slow; this test says so before the GPU run does."""
import copy
import json
import os

import torch

import test_gpu_epoch_parity as T


def test_fp64_envelope_fixture_matches_its_inputs():
    with open(T._FX_PATH) as f:
        fx = json.load(f)
    lr = 1e-4
    want = {}
    for style in ("partial", "simple"):
        cohort = T._cohort()
        ref64 = copy.deepcopy(T._ref(T.STYLES[style][0], 11)).double()
        want["epoch-%s-%g" % (style, lr)] = T._fingerprint(ref64, [cohort["image"], cohort["rnaseq"], cohort["label"]], T._hyper(style, lr))
    cohort = T._cohort()
    for f_, (seed, split) in enumerate(zip((21, 22), T.LOCKSTEP_SPLITS)):
        ref64 = copy.deepcopy(T._ref("PartialModalityNet", seed)).double()
        want["lockstep-partial-f%d-%g" % (f_, lr)] = T._fingerprint(ref64, [cohort["image"], cohort["rnaseq"], cohort["label"], torch.as_tensor(split)],
                                                                    T._hyper("partial", lr))
    cohort, tr, va = T._config1_inputs(lr)
    ref64 = copy.deepcopy(T._ref("SimpleFusionModel", 31, rna_dim=5005)).double()
    want["config1-%g" % lr] = T._fingerprint(ref64, [cohort["rnaseq"], cohort["label"], torch.as_tensor(tr)], T._hyper("simple", lr, extra="config1"))
    assert set(fx) == set(want), (sorted(fx), sorted(want))
    for k, fp in want.items():
        assert fx[k]["fingerprint"] == fp, (k, fx[k]["fingerprint"], fp, "regenerate: python tests/golden/generate_fp64_envelope.py")
    # shape of the stored values
    assert len(fx["epoch-partial-0.0001"]["value"]["pb64"]) == 6 and len(fx["epoch-partial-0.0001"]["value"]["want64"]) == 2
    assert len(fx["epoch-simple-0.0001"]["value"]["pb64"]) == 4 and len(fx["epoch-simple-0.0001"]["value"]["want64"]) == 1
    assert os.path.exists(os.path.join(os.path.dirname(T._FX_PATH), "generate_fp64_envelope.py"))
