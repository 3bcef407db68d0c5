"""Drop-in boundary on the real thing: the reference's own configs/inference_512_v1.0.yaml values
(tests/golden/inference_512_v1.0.model.yaml) instantiated through `instantiate_from_config` on the HIP backend -- with the
PyTorch-ROCm custom-op binding north_star names (TORCH_LIBRARY(tooncrafter), the default) and with the ctypes
binding (TC_BINDING=ctypes) --, a strict load of a synthetic checkpoint in the reference's format, and `image_guided_synthesis` for
2 DDIM steps at 320 x 512 x 16 frames (reference scripts/evaluation/inference.py:180-277, 280-344).
The work is done by tests/real_yaml_run.py in a subprocess (the binding is chosen when the backend is created)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("binding", ["torch", "ctypes"])
def test_real_yaml_end_to_end_on_hip(binding):
    env = dict(os.environ, TC_BINDING=binding)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "real_yaml_run.py")], capture_output=True, text=True,
                       env=env, timeout=1400)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    print(d)
    assert d["binding"] == ("TorchLibOps" if binding == "torch" else "HipOps")
    assert d["shape"] == [1, 1, 3, 16, 320, 512] and d["finite"] and d["std"] > 1e-3
    assert d["state_dict_keys"] > 2500 and d["params_B"] > 2.4           # UNet + AE + both OpenCLIP towers + Resampler
