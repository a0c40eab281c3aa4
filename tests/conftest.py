import gc
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a ROCm device (run on the MI355X box with -m gpu)')


@pytest.fixture(scope='session')
def golden_dir():
    return os.path.join(ROOT, 'tests', 'golden')


@pytest.fixture(autouse=True)
def _gpu_test_isolation(request):
    """Every GPU test ends with a drained device and the configuration it started with.

    An asynchronous GPU fault (a kernel reading through a bad pointer) kills the process at the NEXT runtime call: without
    the synchronisation below that is some later test's first line, and the fault cannot be attributed (the round-4 GPUTEST
    abort was reported against ``test_hipgraph_replay_equals_eager`` and belonged to the overflow test in front of it).
    After the test: wait for the device, collect garbage (graph objects release their captures when they die), wait again, and put ``exa.config`` back -- tests flip its process-global knobs.
    ``EXA_TEST_POISON=1`` additionally fills every rasterizer workspace with 0xFF before it is used (``config.poison``)."""
    if request.node.get_closest_marker('gpu') is None:
        yield
        return
    import torch
    import exavatar_release_amd as exa
    cfg = exa.config
    saved = {k: getattr(cfg, k) for k in dir(cfg) if not k.startswith('_')}
    if os.environ.get('EXA_TEST_POISON'):
        cfg.poison = True
    yield
    try:
        if torch.cuda.is_available():
            torch.cuda.synchronize()
            gc.collect()
            torch.cuda.synchronize()
    finally:
        for k, v in saved.items():
            setattr(cfg, k, v)
        if os.environ.get('EXA_TEST_POISON'):
            cfg.poison = True


def pytest_sessionfinish(session, exitstatus):
    """GPU runs: how many renders the compiled autograd node took (gpurun_out/compiled_node_calls.txt, when that directory exists)."""
    import sys as _sys
    rz = _sys.modules.get('exavatar_release_amd.rasterizer')
    out = os.path.join(ROOT, 'gpurun_out')
    if rz is not None and getattr(rz, 'compiled_calls', 0) and os.path.isdir(out):
        with open(os.path.join(out, 'compiled_node_calls.txt'), 'w') as f:
            f.write('%d renders of this pytest session went through the compiled autograd node (_exa_torch)\n' % rz.compiled_calls)
