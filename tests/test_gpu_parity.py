"""GPU parity (run with -m gpu on an MI355X): the HIP library, called through the C ABI, against
  (1) the golden traces captured from the reference, and
  (2) the CPU oracle on seeded random batches, step for step.
Integer outputs bit-exact; fp64 state bit-exact as well (tolerance stated in replay.FTOL = 0)."""
import numpy as np
import pytest
import torch

from replay import ALL_TRACES, Replay, TRACES_NOMOVE

pytestmark = pytest.mark.gpu


def pkg_abi_version():
    import drone2d_amd
    return drone2d_amd._abi.D2D_ABI_VERSION


def test_native_library_is_loaded(hip):
    assert hip.fn['abi_version']() == pkg_abi_version()
    import drone2d_amd
    with open('/proc/self/maps') as f:
        assert 'libd2d_hip.so' in f.read()


def test_device_tan_matches_host_libm(hip, oracle):
    """d2d_tan on the device == the libm tan Python's math.tan calls, bit for bit."""
    rng = np.random.RandomState(7)
    xs = [rng.uniform(0, 2 * np.pi, 4_000_000), rng.uniform(-25, 25, 1_000_000), rng.uniform(-0.07, 0.07, 500_000),
          rng.uniform(-1e-7, 1e-7, 100_000)]
    for k in range(9):   # +-2000 ulp around every multiple of pi/4 (the structured ray angles)
        c = k * np.pi / 4
        v = np.full(4001, c)
        for j in range(2000):
            v[:2000 - j] = np.nextafter(v[:2000 - j], -np.inf)
            v[2001 + j:] = np.nextafter(v[2001 + j:], np.inf)
        xs.append(v)
    x = np.concatenate(xs)
    xh = torch.from_numpy(x)
    ref = torch.empty_like(xh)
    oracle.tan_array(xh, ref)
    xd = xh.to(hip.device)
    out = torch.empty_like(xd)
    hip.tan_array(xd, out)
    hip.sync()
    a = out.cpu().numpy().view(np.int64)
    b = ref.numpy().view(np.int64)
    assert np.array_equal(a, b), f'{(a != b).sum()} of {len(a)} tan values differ'


@pytest.mark.parametrize('name', ALL_TRACES)
def test_hip_replays_reference_trace(pkg, hip, name):
    Replay(pkg, hip, name, copies=5).run(mode='fused', envs=(0, 4))


@pytest.mark.parametrize('name', ['nomove_n10_rand_map2', 'readme_oxford_primitive', 'nomove_big_map'])
@pytest.mark.parametrize('mode', ['split', 'stages'])
def test_hip_split_entry_points(pkg, hip, name, mode):
    Replay(pkg, hip, name, copies=3).run(mode=mode, envs=(0, 2))


def test_hip_replayed_active_bits(pkg, hip):
    Replay(pkg, hip, 'nomove_n10_rand_map3', kf=False, copies=2).run(mode='fused', envs=(1,))
