"""CPU harness check only (tests/test_reference_scripts_cpu.py): the reference's scripts hard-code ``.cuda()``; with this
directory on PYTHONPATH the call is the identity, so the reference's OWN modules can run them on the host -- which is how
the stubs, the synthetic data and the script plumbing are validated where there is no GPU.  Never on the product path."""
import torch

torch.Tensor.cuda = lambda self, *a, **k: self
torch.nn.Module.cuda = lambda self, *a, **k: self
