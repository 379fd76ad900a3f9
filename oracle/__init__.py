"""CPU oracle for the reverse-diffusion sampling path — TEST INFRASTRUCTURE ONLY.

This package restates, on the CPU, the algorithm of the reference's hot path
(`/root/reference/m_diffuser/{models/temporal_unet.py, models/diffusion.py,
guides/policies.py, dynamics/projection.py}`), function by function with file:line
citations.  It exists to CHECK the HIP path; it is never the thing shipped or measured.

Who may import it: ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg
of ``bench.py``.  Nothing under ``dynamics_aware_diffusion_amd/`` imports it, and the
product path raises when the HIP library is missing instead of falling back to this.

Parity status: PINNED.  Every function here is compared (``tests/test_oracle_golden.py``)
against golden vectors produced by importing the real reference modules in the build
container (``tests/golden/make_golden.py``).  The reference's own test
suite holds no golden vectors for this path.

Arithmetic: the reference is stock ``torch.nn`` in fp32 (``requirements.txt:2`` torch>=2.0,
unpinned; fixtures made with torch 2.10.0 CPU).  The restatement calls the same ATen CPU
primitives through ``torch.nn.functional`` — so it doubles as the "reference --device cpu"
timing stand-in on the GPU box, where the reference itself never travels — and can be run
in float64 (``dtype=torch.float64``) to provide a higher-precision truth.
"""
