"""Streamed attention forward with parts switched off (VSDE_AS_ABLATE bits; results are wrong, only the time matters)."""
import os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
for bits in (0, 1, 2, 4, 8, 16, 24, 3, 6, 7, 31):
    env = dict(os.environ, VSDE_AS_ABLATE=str(bits), VSDE_AS_FWD_ONLY="1")
    out = subprocess.run([sys.executable, os.path.join(here, "attn_stream_bench.py")], env=env, capture_output=True, text=True).stdout
    print(bits, out.strip().splitlines()[0] if out.strip() else "failed")
