import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hand():
    from myosuite_mjx_amd import model as M
    return M.load_asset("myohand_pose")


@pytest.fixture(scope="session")
def finger():
    from myosuite_mjx_amd import model as M
    return M.load_asset("myofinger_v0")


@pytest.fixture(scope="session")
def legs():
    from myosuite_mjx_amd import model as M
    return M.load_asset("myolegs")


@pytest.fixture(scope="session")
def elbow():
    from myosuite_mjx_amd import model as M
    return M.load_asset("myoelbow_1dof6muscles")


@pytest.fixture(scope="session")
def exo():
    from myosuite_mjx_amd import model as M
    return M.load_asset("myoelbow_1dof6muscles_1dofexo")


@pytest.fixture(scope="session")
def motorfinger():
    from myosuite_mjx_amd import model as M
    return M.load_asset("motorfinger_v0")


@pytest.fixture(scope="session")
def terrain():
    from myosuite_mjx_amd import model as M
    return M.load_asset("myolegs_terrain")


@pytest.fixture(scope="session")
def oracle64(hand):
    from oracle.oracle import Oracle
    return Oracle(hand.blob())


@pytest.fixture(scope="session")
def oracle32(hand):
    from oracle.oracle import Oracle
    return Oracle(hand.blob(), f32=True)


REFERENCE = "/root/reference/myosuite"
needs_reference = pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="reference tree not present (GPU box)")


@pytest.fixture(scope="session")
def legoracle64(legs):
    from oracle.oracle import Oracle
    return Oracle(legs.blob())
