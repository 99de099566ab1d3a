"""The library must bind to the HIP runtime PyTorch ships, whatever is imported first (a process
with two runtimes sees no device from the second one): build() followed by smoke() in one
process is exactly the "library first" order."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize('code', [
	'from nway_amd import _hip; _hip.load(); import torch; assert torch.cuda.is_available() and _hip.device_count() >= 1',
	'import torch; from nway_amd import _hip; _hip.load(); assert torch.cuda.is_available() and _hip.device_count() >= 1',
	'import __graft_entry__ as g; g.build(); g.smoke()',
])
def test_library_and_torch_share_one_runtime(code):
	res = subprocess.run([sys.executable, '-c', code], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
		universal_newlines=True, timeout=600)
	assert res.returncode == 0, res.stdout[-2000:]
