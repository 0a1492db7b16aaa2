"""The bench line contract, checked on the committed line of the latest measured build (profiles/) -- no GPU needed."""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def latest_line():
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_bench_v*_final.json')))
    assert files, 'no committed bench line under profiles/'
    return json.load(open(files[-1])), files[-1]


def test_committed_bench_line_has_the_contract_fields():
    d, path = latest_line()
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert key in d, '%s: missing %r' % (path, key)
    assert d['n_gpus'] == 1 and d['higher_is_better'] is True and d['dtype'] == 'f64' and d['data'] == 'synthetic'
    assert d['vs_baseline'] is None                       # BASELINE.json publishes no number
    assert 'workload' in d['config'] and 'model' not in d['config']
    assert abs(d['value'] * d['ms_per_step'] * 1e-3 - 512 * 512 * 256 * 512) <= 1e-6 * 512 * 512 * 256 * 512   # cells x steps x chains of one fit
    r = d['roofline']
    assert r['bound'] == 'hbm' and r['unit'] == 'GB/s' and r['peak'] == 8000.0
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-12 and 0.0 < r['frac'] < 1.0
    assert abs(r['achieved'] - r['bytes_per_cell_step'] * r['cells_per_launch'] / (r['avg_launch_us'] * 1e-6) / 1e9) < 1e-6 * r['achieved']
    # HBM bytes per launch (PMC): a resident kernel moves LESS than the streaming formulation's algorithmic bytes (C4 backward + fold:
    # 24 of 32 B per cell-step), never much more
    alg = r['bytes_per_cell_step'] * r['cells_per_launch']
    assert r['traffic'] is None or 0.2 * alg <= r['traffic'] <= 1.3 * alg
    c = d['cpu_baseline']
    assert c['kind'] == 'port' and c['cores'] >= 1 and c['unit'] == d['unit'] and c['value'] > 0 and c['sample']
