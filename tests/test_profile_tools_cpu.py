"""The profile folding tools (tools/rocprof_summary.py, profile_tables.py, pmc_summary.py) on a synthetic rocpd-shaped sqlite database:
whole-step windows delimited by the once-per-step marker kernel, initialisation kernels and partial steps left out, names that
collide after shortening summed, one marker row per dispatch even when a counter has several rows per dispatch."""
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))


def _fake_db(path, steps=4, counter=None, rows_per_dispatch=1):
    db = sqlite3.connect(path)
    cur = db.cursor()
    if counter is None:
        cur.execute('create table kernels (name text, start integer, end integer, grid_x integer, workgroup_x integer)')
    else:
        cur.execute('create table pmc_events (name text, start integer, end integer, counter_name text, counter_value real, dispatch_id integer)')
    t, did = 1000, 0

    def launch(name, dur, grid=256 * 198, val=0.0):
        nonlocal t, did
        did += 1
        if counter is None:
            cur.execute('insert into kernels values (?,?,?,?,?)', (name, t, t + dur, grid, 256))
        else:
            for _ in range(rows_per_dispatch):
                cur.execute('insert into pmc_events values (?,?,?,?,?,?)', (name, t, t + dur, counter, val / rows_per_dispatch, did))
        t += dur + 10

    for _ in range(50):
        launch('init_fill_kernel(float*)', 100, val=1e6)               # initialisation: must not be counted
    for _ in range(steps):
        launch('void (anonymous namespace)::patchify_kernel(int)', 50, val=10.0)
        launch('void phg::big::gemm_big_kernel<4, false, true>(phg::GemmParams)', 400, val=1000.0)
        launch('void (anonymous namespace)::gemm_ks2_kernel<false, true>(phg::GemmParams)', 100, grid=512 * 180, val=100.0)
        launch('void (anonymous namespace)::ce_fwd_kernel(int)', 30, val=1.0)
        launch('void (anonymous namespace)::adamw_kernel(float*)', 200, val=500.0)
    db.commit(); db.close()


def test_step_window_and_tables(tmp_path):
    import profile_tables as pt
    kt = str(tmp_path / 'kt.db')
    _fake_db(kt, steps=5)
    cur = sqlite3.connect(kt).cursor()
    t0, t1 = pt.step_window(cur, 3)
    assert t1 > t0
    dur = pt.durations(kt, 3)
    wall = dur.pop('__wall_ms_per_step__')[1]
    assert abs(wall - (50 + 400 + 100 + 30 + 200 + 50) / 1e6) < 1e-9
    assert dur['big::gemm_big_kernel<4, false, true>'] == (1.0, 0.4) and 'init_fill_kernel' not in dur
    for c, rows in (('FETCH_SIZE', 1), ('SQ_VALU_MFMA_BUSY_CYCLES', 8)):
        f = str(tmp_path / f'{c}.db')
        _fake_db(f, steps=3, counter=c, rows_per_dispatch=rows)
        got = pt.counter(f, c, 3)
        assert abs(got['big::gemm_big_kernel<4, false, true>'][0] - 1000.0) < 1e-6, got          # per whole step, init excluded
        assert 'init_fill_kernel' not in got


def test_pmc_summary_and_rocprof_summary_cli(tmp_path):
    f, w, kt = str(tmp_path / 'f.db'), str(tmp_path / 'w.db'), str(tmp_path / 'kt.db')
    _fake_db(f, steps=3, counter='FETCH_SIZE')
    _fake_db(w, steps=3, counter='WRITE_SIZE')
    _fake_db(kt, steps=6)
    out = str(tmp_path / 'pmc.json')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'pmc_summary.py'), f, w, '3', out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    import json
    d = json.load(open(out))
    assert d['whole_steps'] == 2.0 and d['launches_per_step'] == 2.0                     # both GEMM kernels, two whole steps
    per_step_kb = 2 * (10 + 1000 + 100 + 1 + 500) + (10 + 1000 + 100 + 1 + 500)
    assert abs(d['whole_step_hbm_gb'] - per_step_kb * 1024 / 1e9) < 1e-9
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'rocprof_summary.py'), kt, str(tmp_path / 's.csv'), '4', '400'],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert 'over 4 steps' in r.stdout and 'n/step=    1.0' in r.stdout and 'blocks=   198' in r.stdout and 'init_fill' not in r.stdout
