"""Re-run ONE seeded case of tests/test_gpu_random_configs.py with diagnostics: which frames differ from the
oracle's bytes, what the device decoder says about each.  `python3 tools/soak/diag_case.py SEED [SEED ...]`"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import test_gpu_random_configs as T   # noqa: E402
import _oracle as orc                 # noqa: E402
from _compare import orc_options_for, planar_frames   # noqa: E402


def describe(seed):
    """The case's parameters and PCM (the generator of the test, replayed)."""
    c, pcm = T.case_of(seed)
    return c, pcm


def main():
    from flac_codec_amd.gpu import GpuAnalyzer

    for seed in map(int, sys.argv[1:]):
        c, pcm = describe(seed)
        print(seed, c)
        an = GpuAnalyzer(c["block"], c["max_po"], c["max_lpc"], c["mid_side"], c["exhaustive"], c["window"][0],
                         c["window"][1], c["bps"], c["channels"], max_frames=c["n_frames"])
        data, off = an.encode_frames(pcm, c["n_frames"], c["last"], c["first"], c["rate"])
        res, per = an.verify_device(c["rate"], c["first"])
        print(" verify:", {k: getattr(res, k) for k, _ in res._fields_}, per)
        oopts = orc_options_for(c["block"], c["max_po"], c["max_lpc"], c["mid_side"], c["exhaustive"], c["window"][0],
                                c["window"][1])
        for f, planar in enumerate(planar_frames(pcm, c["channels"], c["block"])):
            rc, fb, _ = orc.encode_frame(oopts, c["rate"], c["bps"], planar, frame_number=c["first"] + f)
            g = data[off[f]:off[f + 1]]
            print(f" frame {f}: oracle rc {rc} {len(fb)} bytes, gpu {len(g)} bytes, equal {g == fb}")
            if g != fb:
                print("  oracle:", fb[:48].hex())
                print("  gpu   :", bytes(g[:48]).hex())
        an.close()


if __name__ == "__main__":
    main()
