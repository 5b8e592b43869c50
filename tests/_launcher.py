"""A process that never touches the GPU and starts programs on request (tests/conftest.py starts it before any test of a
`-m gpu` session has initialised HIP in the pytest process: a process that holds the GPU must not fork + exec others).

Protocol: one JSON object per line on stdin {"argv": [...], "env": {...}, "timeout": seconds, "cwd": path}; one JSON object
per line on stdout {"rc": int, "stdout": str, "stderr": str (tail)}.  Ends at EOF.
"""
import json
import os
import subprocess
import sys


def main() -> int:
    for line in sys.stdin:
        line = line.strip()
        if not line:
            continue
        req = json.loads(line)
        env = dict(os.environ)
        env.update(req.get("env") or {})
        try:
            p = subprocess.run(req["argv"], env=env, cwd=req.get("cwd"), capture_output=True, text=True, timeout=req.get("timeout", 600))
            ans = {"rc": p.returncode, "stdout": p.stdout, "stderr": p.stderr[-4000:]}
        except subprocess.TimeoutExpired as exc:
            ans = {"rc": -999, "stdout": (exc.stdout or b"").decode(errors="replace") if isinstance(exc.stdout, bytes) else (exc.stdout or ""),
                   "stderr": "timeout"}
        sys.stdout.write(json.dumps(ans) + "\n")
        sys.stdout.flush()
    return 0


if __name__ == "__main__":
    sys.exit(main())
