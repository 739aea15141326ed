import os
import sys
import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)

# ---- bounds-checked pass (SPI_EFENCE=1, tools/efence_pytest.sh): every torch device allocation becomes its own mapping between unmapped guard
# pages (tools/efence/efence_alloc.cpp), so an out-of-range access of any kernel faults inside the test that made it.  One tensor per
# mapping needs the per-iteration accumulator arena and the HIP-graph memory pools off (both carve many tensors out of one allocation).
EFENCE = os.environ.get('SPI_EFENCE') == '1'
if EFENCE:
    os.environ['SPI_ZERO_ARENA'] = '0'
    os.environ['DEBUG_CLR_GRAPH_PACKET_CAPTURE'] = '1'           # -> spi_amd.hip_graphs_safe() is False: eager iterations, no capture-time allocations
    _so = os.path.join(ROOT, 'tools', 'efence', 'libefence.so')
    assert os.path.exists(_so), f'{_so} missing: hipcc -O2 -shared -fPIC -o {_so} tools/efence/efence_alloc.cpp'
    torch.cuda.memory.change_current_allocator(torch.cuda.memory.CUDAPluggableAllocator(_so, 'efence_malloc', 'efence_free'))

import spi_amd  # noqa: E402,F401  (before the first GPU call: the HIP runtime switch for graph replays, spi_amd/__init__.py)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _restore_config_modules():
    """The reference keeps its run configuration in module globals (configs/global_config.py, hyperparameters.py, paths_config.py) and
    run_inversion.parse_args writes into them; so do several tests.  Snapshot the three modules before every test and put them back
    afterwards, so the suite does not depend on the order the tests run in (round-2 advisor finding)."""
    from spi_amd.configs import global_config, hyperparameters, paths_config
    mods = (global_config, hyperparameters, paths_config)
    saved = [{k: v for k, v in vars(m).items() if not k.startswith('__')} for m in mods]
    yield
    for m, old in zip(mods, saved):
        for k in [k for k in vars(m) if not k.startswith('__') and k not in old]:
            delattr(m, k)
        for k, v in old.items():
            setattr(m, k, v)


class Golden:
    def __init__(self, name):
        self.z = np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)

    def __getitem__(self, k):
        return torch.from_numpy(np.asarray(self.z[k]))

    def __contains__(self, k):
        return k in self.z.files

    def keys(self):
        return self.z.files


@pytest.fixture(scope='session')
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]
    return get


def rel_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def assert_close(a, b, rel=1e-5, what=''):
    e = rel_err(a, b)
    assert e <= rel, f'{what}: max-normalised error {e:.3e} > {rel:.1e}'


def assert_close_elementwise(a, b, rel=1e-3, floor=1e-2, what=''):
    """Element-wise form of the north star's "1e-3 rel on rendered RGB / depth" (round-5 review, weak #2): EVERY element within
    rel * (|b| + floor * max|b|) of the reference -- a relative bound per pixel with an absolute floor of 1 % of the tensor's range for the pixels near
    zero (an image in [-1, 1] has many).  The max-normalised `assert_close` would pass a kernel that is wrong on small-magnitude pixels; this does not."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    bound = rel * (b.abs() + floor * b.abs().max())
    bad = (a - b).abs() > bound
    share = bad.double().mean().item()
    worst = ((a - b).abs() / bound).max().item()
    assert share == 0.0, f'{what}: {share:.3e} of the elements off by more than {rel:.0e} * (|ref| + {floor:.0e} max|ref|); worst {worst:.2f} x the bound'
    return worst
