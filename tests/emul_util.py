"""TEST INFRASTRUCTURE: load the host-emulated build of the HIP kernels (tests/emul) and inject it into
``dprox._backend`` so the unchanged host layer + kernel sources can be exercised on CPU tensors."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMUL = os.path.join(ROOT, "tests", "emul")


def build():
    subprocess.run([os.path.join(EMUL, "build_emul.sh")], check=True, capture_output=True)
    return os.path.join(EMUL, "libdpx_emul.so")


def use_emulator():
    from dprox import _backend as be, _ops
    if not be.host_mode():
        be._inject_for_tests(be.Library(build()), host_pointers=True)
        _ops.clear_caches()
    return be
