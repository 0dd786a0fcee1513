"""CPU: bench.py's own launcher (`python bench.py --gpus N` without torchrun) -- the parent starts N rank processes with RANK /
LOCAL_RANK / WORLD_SIZE / MASTER_* set, relays rank 0's stdout and fails when any rank fails -- exercised with a stand-in child
(the real children need GPUs: tests/test_gpu_ddp.py runs the launcher with one on the GPU box)."""
import os
import subprocess
import sys
import textwrap

from conftest import ROOT


def _run_launcher(tmp_path, child_body, n=3, extra=()):
    # a copy of bench.py whose main() is replaced by the stand-in once WORLD_SIZE is set (= in the children)
    src = open(os.path.join(ROOT, 'bench.py')).read()
    marker = "    rank = int(os.environ.get('RANK', 0))\n"
    assert marker in src
    src = src.replace(marker, textwrap.indent(textwrap.dedent(child_body), '    ') + '    return\n' + marker, 1)
    path = os.path.join(tmp_path, 'bench_standin.py')
    with open(path, 'w') as f:
        f.write(src.replace("ROOT = os.path.dirname(os.path.abspath(__file__))", f"ROOT = {ROOT!r}"))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE')}
    return subprocess.run([sys.executable, path, '--gpus', str(n), '--steps', '2', *extra], env=env, capture_output=True, text=True, timeout=300)


def test_launcher_starts_n_ranks_and_relays_rank0(tmp_path):
    r = _run_launcher(tmp_path, '''
        import json
        rk = int(os.environ['RANK'])
        assert os.environ['LOCAL_RANK'] == os.environ['RANK'] and os.environ['MASTER_ADDR'] == '127.0.0.1' and int(os.environ['MASTER_PORT']) > 0
        open(os.path.join(os.path.dirname(os.path.abspath(__file__)), f'rank{rk}.seen'), 'w').write(os.environ['WORLD_SIZE'])
        print(json.dumps(dict(rank=rk, world=int(os.environ['WORLD_SIZE']), gpus=args.gpus, steps=args.steps)), flush=True)
    ''')
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert lines == ['{"rank": 0, "world": 3, "gpus": 3, "steps": 2}']       # ONE line: rank 0's
    assert sorted(f for f in os.listdir(tmp_path) if f.endswith('.seen')) == ['rank0.seen', 'rank1.seen', 'rank2.seen']


def test_launcher_hands_the_masked_objective_to_every_rank(tmp_path):
    """`python bench.py --gpus N --objective masked` (the north_star's data-parallel loop is the PRE-TRAIN step): the ranks see the objective
    and the rest of the command line unchanged"""
    r = _run_launcher(tmp_path, '''
        import json
        print(json.dumps(dict(rank=int(os.environ['RANK']), objective=args.objective, gpus=args.gpus, batch=args.batch)), flush=True)
    ''', n=2, extra=('--objective', 'masked', '--batch', '64'))
    assert r.returncode == 0, r.stderr
    assert [l for l in r.stdout.splitlines() if l.startswith('{')] == ['{"rank": 0, "objective": "masked", "gpus": 2, "batch": 64}']


def test_launcher_fails_when_a_rank_fails(tmp_path):
    r = _run_launcher(tmp_path, '''
        if os.environ['RANK'] == '1':
            sys.exit(7)
        print('{"ok": true}', flush=True)
    ''')
    assert r.returncode != 0 and 'ranks failed' in r.stderr and '(1, 7)' in r.stderr


def test_launcher_watchdog_ends_the_surviving_ranks(tmp_path):
    """a rank that dies after the others have started (they sit in a rendezvous / collective, here: a long sleep) takes the job down
    within seconds instead of leaving them to a 10-minute collective timeout"""
    import time
    t0 = time.monotonic()
    r = _run_launcher(tmp_path, '''
        import time as _t
        rk = int(os.environ['RANK'])
        open(os.path.join(os.path.dirname(os.path.abspath(__file__)), f'rank{rk}.started'), 'w').write('x')
        if rk == 2:
            while len([f for f in os.listdir(os.path.dirname(os.path.abspath(__file__))) if f.endswith('.started')]) < 3:
                _t.sleep(0.05)           # die only once every peer is up
            sys.exit(9)
        print('{"partial": true}', flush=True)
        _t.sleep(600)                    # stand-in for a rank blocked in a collective
    ''')
    dt = time.monotonic() - t0
    assert r.returncode != 0 and '(2, 9)' in r.stderr and 'failed first' in r.stderr
    assert dt < 60, dt
    assert sorted(f for f in os.listdir(tmp_path) if f.endswith('.started')) == ['rank0.started', 'rank1.started', 'rank2.started']


def test_world_size_mismatch_is_an_error_not_an_assert(tmp_path):
    env = dict(os.environ, WORLD_SIZE='2', RANK='0', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '4'], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and 'WORLD_SIZE=2' in r.stderr and 'AssertionError' not in r.stderr


def test_binding_has_no_process_global_launch_state():
    import ecg_representation_learning_amd as E
    assert not hasattr(E.hip, 'GEMM_TILES_PER_WORKGROUP')
    env = dict(os.environ, ECGVIT_HIP_LIB='/nonexistent/other.so')
    r = subprocess.run([sys.executable, '-c', 'import sys; sys.path.insert(0, %r); import ecg_representation_learning_amd as E; print(E.hip.LIB_PATH)' % ROOT],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith(os.path.join('ecg-representation-learning_amd', 'libecgvit_hip.so'))   # the env var is ignored
