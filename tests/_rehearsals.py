"""Supervisor of the child processes a `-m gpu` pytest session needs (started by tests/conftest.py BEFORE the pytest process
touches the GPU: on the GPU boxes a process that has initialised the GPU must not start another program).  It runs the jobs of a
JSON spec ONE AFTER THE OTHER — each a fresh `python ...` process with its own environment — and leaves, per job, `<name>.out`,
`<name>.err` and, last, `<name>.rc` in the output directory; the tests poll for the `.rc` file.  This process never imports torch
and never touches the GPU itself.

spec: {"dir": "...", "jobs": [{"name": "...", "argv": [...], "env": {...}, "timeout": seconds, "free_port_arg": "--master-port"}]}
free_port_arg: when given, a free TCP port (chosen right before the job starts) is inserted behind that argument."""
import json
import os
import socket
import subprocess
import sys


def _free_port():
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def main():
    spec = json.load(open(sys.argv[1]))
    out_dir = spec["dir"]
    for job in spec["jobs"]:
        name = job["name"]
        argv = list(job["argv"])
        if job.get("free_port_arg"):
            i = argv.index(job["free_port_arg"])
            argv.insert(i + 1, str(_free_port()))
        env = dict(os.environ)
        env.update(job.get("env") or {})
        for k in job.get("unset_env") or ():
            env.pop(k, None)
        rc = 125
        with open(os.path.join(out_dir, name + ".out"), "w") as fo, open(os.path.join(out_dir, name + ".err"), "w") as fe:
            try:
                rc = subprocess.run(argv, env=env, stdout=fo, stderr=fe, cwd=spec.get("cwd") or None,
                                    timeout=float(job.get("timeout", 900))).returncode
            except subprocess.TimeoutExpired:
                rc = 124
            except Exception as e:        # noqa: BLE001 - recorded for the collecting test
                fe.write(f"\n[rehearsal supervisor] {e!r}\n")
        tmp = os.path.join(out_dir, name + ".rc.tmp")
        with open(tmp, "w") as f:
            f.write(str(rc))
        os.replace(tmp, os.path.join(out_dir, name + ".rc"))


if __name__ == "__main__":
    main()
