"""bench.py --scaling strong (SURVEY.md 8(d) config 4's shape: ONE stream cut into contiguous frame ranges over the
ranks): two ranks sharing GPU 0 with gloo collectives (FLAC_BENCH_SHARE_DEVICE=1, so that it runs inside a 1-GPU
lease) must produce, frame ranges gathered and metadata rebuilt on rank 0, the very .flac one writer produces.
/root/reference/src/encode.rs:1999-2003, 2414-2436."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("config,frames", [(3, 22), (4, 7)])
def test_strong_scaling_two_ranks_emit_the_single_writers_stream(tmp_path, config, frames):
    sys.path.insert(0, ROOT)
    import bench
    import _oracle as orc
    from flac_codec_amd.encode import FlacSampleWriter, Options

    out = tmp_path / "strong.flac"
    env = dict(os.environ, FLAC_BENCH_SHARE_DEVICE="1")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--scaling", "strong", "--config",
                        str(config), "--frames", str(frames), "--steps", "2", "--warmup", "1", "--prewarm-ms", "0",
                        "--sustained-steps", "0", "--contexts", "2", "--no-cpu-baseline", "--no-end-to-end",
                        "--no-other-configs", "--emit-flac", str(out), "--detail", str(tmp_path / "detail.json")],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["scaling"] == "strong" and line["n_gpus"] == 2
    assert line["config"]["frames_per_step"] == frames and line["config"]["frames_per_gpu"] == frames // 2
    assert line["extra"]["shards"]["total_frames"] == frames and line["extra"]["shards"]["ranks_seen"] == 2
    assert len(r.stdout.strip().splitlines()[-1]) < 4096          # the compact record the driver parses
    detail = json.load(open(tmp_path / "detail.json"))           # ... and the full one beside it
    assert detail["shard_counters"]["total_frames"] == frames and detail["shard_counters"]["backend"] == "gloo"
    cfg = bench.CONFIGS[config]
    pcm = bench.make_pcm(1000 + 16 * config, frames, cfg["ch"], cfg["bps"])
    o = Options.best() if cfg["lpc"] >= 12 else Options.default()
    o = o.max_lpc_order(cfg["lpc"] or None).max_partition_order(cfg["po"])
    w = FlacSampleWriter(None, o, cfg["rate"], cfg["bps"], cfg["ch"], pcm.size)
    w.write(pcm)
    w.finalize()
    single = w.getvalue()
    w.close()
    data = out.read_bytes()
    assert data == single, "the strong-scaling run's stream differs from the single writer's"
    rc, ref, _ = orc.encode_stream(bench.orc_options(orc, cfg), cfg["rate"], cfg["bps"], cfg["ch"], pcm, total_known=True)
    assert rc == 0 and data == ref
    rc, dec, info = orc.decode_stream(data)
    assert rc == 0 and info.md5_ok == 1 and np.array_equal(dec, pcm)


def test_in_process_multi_device_bench_line():
    """bench.py --gpus 2 --in-process: ONE process, two shards (both on GPU 0 here) through flacgpu_multi_*: the compact
    record carries the whole-job rate, the merged counters and a roofline; every shard's frames are oracle-checked."""
    env = dict(os.environ, FLAC_BENCH_SHARE_DEVICE="1")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--in-process", "--frames", "64",
                        "--steps", "3", "--warmup", "1", "--prewarm-ms", "0", "--contexts", "2", "--detail", os.devnull],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["config"]["frames_per_step"] == 128 and line["scaling"] == "weak"
    sh = line["extra"]["shards"]
    assert sh["ranks_seen"] == 2 and sh["total_frames"] == 128 and sh["min_frame"] <= sh["max_frame"]
    assert line["extra"]["parity"]["identical_per_rank"] == 64 and line["roofline"]["kernel"]
    assert abs(line["value"] - 128 * 4096 * 2 / (line["ms_per_step"] * 1e-3) / 1e6) / line["value"] < 1e-3
