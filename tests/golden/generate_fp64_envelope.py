"""Regenerates tests/golden/fp64_envelope.json: the fp64 runs of the ORACLE loops (oracle/loops.py on the CPU, no HIP code involved)
that tests/test_gpu_epoch_parity.py uses as the exact reference of its fp64-envelope criteria -- per-batch losses and epoch means of
train_epoch_partial / train_epoch_simple on the 22-patient special-case epoch (lr = 1e-4), and BASELINE config 1's train + validate
values on its first 8 batches.  CPU only (a few minutes):  python tests/golden/generate_fp64_envelope.py
Each entry carries a fingerprint of its inputs (sha256 of the initial weights, the cohort tensors, the hyper-parameters and the text of
oracle/*.py, + the torch version); a test whose inputs differ ignores the
fixture and computes the leg live (MMS_FP64_LIVE=1 forces that)."""
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["MMS_WRITE_FP64_FIXTURES"] = "1"
os.environ["MMS_FP64_LIVE"] = "1"

import torch  # noqa: E402
import test_gpu_epoch_parity as T  # noqa: E402


def main():
    lr = 1e-4
    for style in ("partial", "simple"):
        cohort = T._cohort()
        ref64 = copy.deepcopy(T._ref(T.STYLES[style][0], 11)).double()
        fp = T._fingerprint(ref64, [cohort["image"], cohort["rnaseq"], cohort["label"]], T._hyper(style, lr))
        v = T._fp64_leg("epoch-%s-%g" % (style, lr), fp, lambda: T._fp64_epoch_leg(style, lr, ref64, cohort))
        print(style, v["want64"], flush=True)
    cohort = T._cohort()
    for f_, (seed, split) in enumerate(zip((21, 22), T.LOCKSTEP_SPLITS)):         # test_lockstep_epoch_matches_oracle_loops[1e-4]
        ref64 = copy.deepcopy(T._ref("PartialModalityNet", seed)).double()
        fp = T._fingerprint(ref64, [cohort["image"], cohort["rnaseq"], cohort["label"], torch.as_tensor(split)], T._hyper("partial", lr))
        v = T._fp64_leg("lockstep-partial-f%d-%g" % (f_, lr), fp, lambda: T._fp64_epoch_leg("partial", lr, ref64, cohort, split=split, buffers=True))
        print("lockstep fold", f_, v["want64"], flush=True)
    cohort, tr, va = T._config1_inputs(lr)
    ref64 = copy.deepcopy(T._ref("SimpleFusionModel", 31, rna_dim=5005)).double()
    fp = T._fingerprint(ref64, [cohort["rnaseq"], cohort["label"], torch.as_tensor(tr)], T._hyper("simple", lr, extra="config1"))
    v = T._fp64_leg("config1-%g" % lr, fp, lambda: T._fp64_config1_leg(lr, ref64, cohort, tr, va))
    print("config1", v, flush=True)


if __name__ == "__main__":
    main()
