"""Pins the CPU oracle (oracle/d2d_oracle.c) to the reference: every golden trace captured from the
imported reference Python is replayed through the oracle."""
import pytest

from replay import ALL_TRACES, Replay


@pytest.mark.parametrize('name', ALL_TRACES)
def test_oracle_replays_reference_trace(pkg, oracle, name):
    Replay(pkg, oracle, name).run(mode='fused')


@pytest.mark.parametrize('name', ['nomove_n10_rand_map2', 'readme_oxford_primitive'])
@pytest.mark.parametrize('mode', ['split', 'stages'])
def test_oracle_split_entry_points(pkg, oracle, name, mode):
    Replay(pkg, oracle, name).run(mode=mode)


def test_oracle_replayed_active_bits(pkg, oracle):
    """kf disabled: tracker `active` bits are inputs."""
    Replay(pkg, oracle, 'nomove_n10_rand_map3', kf=False).run(mode='fused')
