"""bench.py prints ONE compact JSON line the driver can parse from an 8 KB tail (the 21.8 KB line of round 4 was recorded as `parsed: null`).
The canned result is the full round-4 record (profiles/r04_bench_driver_cmd.json: every table, companion and provenance string)."""
import importlib.util
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def bench():
    spec = importlib.util.spec_from_file_location('sgv_bench_py', os.path.join(ROOT, 'bench.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def canned():
    with open(os.path.join(ROOT, 'profiles', 'r04_bench_driver_cmd.json')) as fh:
        return json.load(fh)


def test_compact_line_is_small_and_complete(bench):
    full = canned()
    assert len(json.dumps(full)) > 20000
    line = bench.compact_line(full, 'bench_detail.json')
    assert '\n' not in line and len(line) < bench.COMPACT_LIMIT <= 6000
    d = json.loads(line)
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert key in d, key
    assert d['value'] == pytest.approx(full['value'], rel=1e-5) and d['ms_per_step'] == pytest.approx(full['ms_per_step'], rel=1e-5)
    r = d['roofline']
    assert r['bound'] == 'mfma' and r['frac'] == pytest.approx(r['achieved'] / r['peak'], rel=1e-4) and r['traffic'] > 0 and r['unit'] == 'TFLOP/s'
    u = d['roofline_upfirdn2d']
    assert u['bound'] == 'hbm' and u['frac'] == pytest.approx(u['achieved'] / u['peak'], rel=1e-4)
    c = d['cpu_baseline']
    assert c['kind'] in ('port', 'reference') and c['cores'] >= 1 and c['value'] > 0 and c['sample']
    assert d['config']['workload'].startswith('FFS 256x256') and 'model' not in d['config']
    assert 'kernels_by_variant' not in d and 'upfirdn2d_by_size' not in d and 'kernels' not in d
    assert d['detail'] == 'bench_detail.json'


def test_compact_line_says_which_step_the_value_measures(bench):
    full = canned()
    full['config']['headline_mode'] = 'captured'
    full['value_eager'] = 600.0
    d = json.loads(bench.compact_line(full, None))
    assert d['config']['headline_mode'] == 'captured' and d['value_eager'] == 600.0      # the same key must not silently mean two things


def test_compact_line_hard_limit_drops_optional_parts_not_contract_keys(bench):
    full = canned()
    full['value_padding'] = 'x' * 20000          # a future scalar that would blow the line up
    full['multi_gpu'] = dict(ms_per_step_by_rank=[1.0] * 8)
    line = bench.compact_line(full, None)
    assert len(line) < bench.COMPACT_LIMIT
    d = json.loads(line)
    assert d['roofline']['frac'] > 0 and d['cpu_baseline']['value'] > 0 and d['value'] > 0


def test_emit_prints_the_compact_line_last_and_writes_the_side_file(bench, tmp_path, capsys, monkeypatch):
    monkeypatch.setattr(bench, 'ROOT', str(tmp_path))
    os.makedirs(tmp_path / 'gpurun_out')
    bench.emit(canned())
    out = capsys.readouterr().out.strip().split('\n')
    assert len(out) == 1 and len(out[0]) < 6000
    assert json.loads(out[0])['detail'] == 'bench_detail.json'
    for path in (tmp_path / 'bench_detail.json', tmp_path / 'gpurun_out' / 'bench_detail.json'):
        assert 'kernels_by_variant' in json.loads(path.read_text())


def test_power_sampler_is_silent_without_a_device():
    """bench.PowerSampler reads the hwmon files of the HIP device's own card; on a box without one it must do nothing -- no exception, no samples."""
    import bench
    s = bench.PowerSampler(0, period=0.001)
    with s:
        pass
    assert s.summary() is None
