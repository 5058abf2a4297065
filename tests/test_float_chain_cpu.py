"""float_chain.h's wavefront form of the reference's float recurrences (likelihood.cpp:120-134, pf.h:255-260), replayed on the
CPU lane by lane (tests/cpp/float_chain_emul.cpp) against the plain sequential loop it must reproduce bit for bit: likelihood-like
terms, deliberate rounding ties, wild magnitudes (binade crossings everywhere, denormals, huge terms), equal terms with a carried-in
sum, sign changes / NaN / infinity (the serial fallback), weight-like terms."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def test_wavefront_chain_equals_serial_loop(tmp_path):
    exe = tmp_path / "float_chain_emul.bin"
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-o", str(exe), os.path.join(HERE, "cpp", "float_chain_emul.cpp")], check=True)
    out = subprocess.run([str(exe), "120"], check=True, capture_output=True, text=True, timeout=600).stdout
    assert "all equal" in out, out
