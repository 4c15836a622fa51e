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
    """The case's parameters (the generator of one_case, replayed)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    block = int(rng.choice([1024, 1152, 2048, 2304, 4096, 4096, 4096, 192, 576, 1000, 4608, 16, 333, 8192]))
    channels = int(rng.choice([1, 2, 2, 2, 3, 4, 6, 8]))
    bps = int(rng.choice([8, 12, 16, 16, 20, 24, 24, 32]))
    max_lpc = min(int(rng.choice([0, 1, 4, 8, 12, 12, 16, 17, 32])), 32)
    tz = (block & -block).bit_length() - 1
    max_po = int(rng.integers(0, 7))
    if min(tz, max_po) > 6:
        max_po = 6
    mid_side = bool(rng.integers(2))
    exhaustive = bool(rng.integers(2))
    window = [(0, 0.0), (1, 0.0), (2, 0.5), (2, 0.25)][int(rng.integers(4))]
    n_frames = int(rng.integers(1, 6))
    last = block if rng.integers(3) else int(rng.integers(1, block + 1))
    if block <= max_lpc:
        max_lpc = 0
    kind = str(rng.choice(["synth", "synth", "synth", "noise", "silence", "sparse", "quiet", "shifted", "sine"]))
    if kind == "shifted" and bps < 12:
        kind = "synth"
    n = (n_frames - 1) * block + last
    pcm = T.make_signal(rng, kind, channels, bps, n)
    rate = int(rng.choice([8000, 44100, 48000, 96000, 192000, 12345]))
    first = int(rng.choice([0, 127, 128, 70000, (1 << 31) - 8]))
    return dict(block=block, channels=channels, bps=bps, max_lpc=max_lpc, max_po=max_po, mid_side=mid_side,
                exhaustive=exhaustive, window=window, n_frames=n_frames, last=last, kind=kind, rate=rate, first=first), pcm


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
