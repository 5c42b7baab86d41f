#!/usr/bin/env python3
"""Pins the reference harness (oracle/_ref/ref_harness, what the golden lists were made with) to the reference's own
command-line program, once: TEST INFRASTRUCTURE, build container only (needs /root/reference; `make -C oracle ref` builds
oracle/_ref/readsb from the reference's sources where they lie, with its own flags and no SDR libraries).

    python oracle/pin_cli.py [scenario ...]          (default: cfg1_df11_df17 mixed_fix1)

Runs  readsb --device-type ifile --ifile <the scenario's IQ> --iformat UC8 --raw --mlat  (plus --no-fix / --aggressive /
--modeac as the scenario asks), reads the `@<12-hex-digit timestamp><frame hex>;` lines it prints (mode_s.c
displayModesMessage: --raw prints only these, --mlat puts the 12 MHz timestamp in front) and compares them, in order, with
(timestamp, frame bytes) of tests/golden/demod_<scenario>.npz.  The program ends when the file does (sdr_ifile.c:216-221; readsb.c:284 calls that an unexpected exit: status 1).
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:] = [p for p in sys.path if os.path.abspath(p or ".") != os.path.join(ROOT, "oracle")]  # oracle/oracle.py would shadow the package
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle.oracle import REC_DTYPE  # noqa: E402
from readsb_amd.synth import generate  # noqa: E402
from scenarios import SCENARIOS  # noqa: E402

CLI = os.path.join(ROOT, "oracle", "_ref", "readsb")


def cli_lines(iq, fmt, nfix, modeac):
    with tempfile.NamedTemporaryFile(suffix=".iq", dir="/tmp") as f:
        iq.tofile(f)
        f.flush()
        cmd = [CLI, "--device-type", "ifile", "--ifile", f.name, "--iformat", fmt, "--raw", "--mlat"]
        cmd += {0: ["--no-fix"], 1: [], 2: ["--aggressive"]}[nfix] + (["--modeac"] if modeac else [])
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    # the end of the file is the reader thread's "unexpected exit" (readsb.c:284: Modes.exit = 2), so the program ends with
    # "Abnormal exit." and status 1 after it has decoded everything; anything else is a failure of the run
    if out.returncode not in (0, 1) or (out.returncode == 1 and "Abnormal exit" not in out.stderr):
        raise SystemExit(f"{' '.join(cmd)} exited with {out.returncode}: {out.stderr[-400:]}")
    return [l.strip() for l in out.stdout.splitlines() if l.startswith("@") and l.rstrip().endswith(";")], " ".join(cmd[:1] + cmd[5:])


def main(names):
    if not os.path.exists(CLI):
        raise SystemExit("oracle/_ref/readsb is missing: make -C oracle ref (build container only)")
    bad = 0
    for name in names:
        sc, kw = SCENARIOS[name]
        if kw.get("dc_filter"):
            raise SystemExit("--dcfilter scenarios are not pinned here")
        lines, cmd = cli_lines(generate(sc), sc.fmt, kw["nfix"], kw["modeac"])
        z = np.load(os.path.join(ROOT, "tests", "golden", f"demod_{name}.npz"))
        gold = z["msgs"].reshape(-1).view(REC_DTYPE)
        want = ["@%012X%s;" % (int(m["ts"]), bytes(m["msg"][: int(m["msgbits"]) // 8]).hex()) for m in gold]
        same = lines == want
        first = next((i for i, (a, b) in enumerate(zip(lines, want)) if a != b), min(len(lines), len(want)))
        print(f"{name}: {len(lines)} lines from `{cmd}`, {len(want)} golden messages: " +
              ("IDENTICAL" if same else f"DIFFER at {first}: {lines[first] if first < len(lines) else None} / {want[first] if first < len(want) else None}"))
        bad += not same
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:] or ["cfg1_df11_df17", "mixed_fix1"]))
