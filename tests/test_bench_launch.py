"""bench.py --gpus N without torchrun: the parent that spawns the ranks must not have loaded the HIP runtime (on this
pool a process that has touched the GPU must not be the one whose children exec), must count GPUs without HIP, and
must refuse to run fewer ranks than asked for."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROBE = r"""
import sys
sys.argv = ['bench.py', '--gpus', '2', '--steps', '1']
import bench
seen = []
class FakePopen:
    def __init__(self, argv, env=None):
        seen.append((open('/proc/self/maps').read(), dict(env), list(argv), 'torch' in sys.modules))
    def wait(self):
        return 0
bench.count_gpus_without_hip = lambda: int(sys.stdin.readline())
try:
    bench.launch_ranks(2, popen=FakePopen)
except SystemExit as e:
    code = e.code
print('CODE', code, 'SPAWNED', len(seen))
for maps, env, argv, torch_loaded in seen:
    assert 'libamdhip64' not in maps and 'libhsa-runtime' not in maps, 'HIP runtime mapped in the launching parent'
    assert not torch_loaded, 'torch imported in the launching parent'
    assert env['WORLD_SIZE'] == '2' and env['MASTER_ADDR'] == '127.0.0.1' and env['LOCAL_RANK'] == env['RANK']
    assert argv[1].endswith('bench.py') and argv[2:] == ['--gpus', '2', '--steps', '1']
print('RANKS', sorted(e['RANK'] for _, e, _, _ in seen))
"""


def _probe(gpus_present):
    r = subprocess.run([sys.executable, "-c", PROBE], cwd=ROOT, input=f"{gpus_present}\n", text=True,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    return r


def test_parent_spawns_ranks_without_touching_hip():
    r = _probe(2)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "CODE 0 SPAWNED 2" in r.stdout and "RANKS ['0', '1']" in r.stdout, r.stdout


def test_fewer_gpus_than_asked_is_refused_loudly():
    r = _probe(1)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "CODE 2 SPAWNED 0" in r.stdout, r.stdout
    assert "refusing to report" in r.stderr


def test_gpu_count_comes_from_kfd_and_visibility(tmp_path, monkeypatch):
    sys.path.insert(0, ROOT)
    import bench

    n = bench.count_gpus_without_hip()
    assert isinstance(n, int) and n >= 0          # this container has no GPU: 0 (no /sys/class/kfd) is fine
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "")
    assert bench.count_gpus_without_hip() == 0
