"""The reference's Qwen2 operator list now lives in the package (dash-infer_amd/ref_graph.py: bench.py and the host runner
build the measured operator list from it); the tests keep importing it under this name."""
from tests.conftest import load_pkg

load_pkg()
from dash_infer_amd.ref_graph import *  # noqa: F401,F403,E402
from dash_infer_amd.ref_graph import qwen2_graph, register_weights, add_graph  # noqa: F401,E402
