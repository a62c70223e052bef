#!/usr/bin/env python3
"""Clean-process launcher for multi-rank GPU tests.

tests/conftest.py starts ONE instance of this script before pytest touches the GPU and talks to it over pipes. It
never imports torch or HIP, so the rank processes it starts are plain children of a process that has not initialised
the GPU (the pytest process has, by the time the tests run, and must not fork/exec programs itself on the GPU boxes).
Protocol: one JSON request per line on stdin -> one JSON reply per line on stdout.
    {"script": path, "nproc": N, "args": [...], "timeout": seconds}
    -> {"rc": [..], "logs": [tail of each rank's combined stdout/stderr], "timed_out": bool}
"""
import json
import os
import socket
import subprocess
import sys
import tempfile
import time


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run(req):
    n = int(req['nproc'])
    port = free_port()
    procs, logs = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(n), LOCAL_RANK=str(r), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        env.update(req.get('env', {}))
        lf = tempfile.TemporaryFile(mode='w+')
        logs.append(lf)
        procs.append(subprocess.Popen([sys.executable, '-u', req['script']] + [str(a) for a in req.get('args', [])],
                                      stdout=lf, stderr=subprocess.STDOUT, env=env))
    deadline = time.time() + float(req.get('timeout', 600))
    timed_out = False
    while any(p.poll() is None for p in procs):
        if time.time() > deadline or any(p.poll() not in (None, 0) for p in procs):
            timed_out = time.time() > deadline
            time.sleep(2.0)                      # a failing rank: give its peers a moment, then stop exactly them
            for p in procs:
                if p.poll() is None:
                    p.kill()
            break
        time.sleep(0.1)
    rc = [p.wait() for p in procs]
    tails = []
    for lf in logs:
        lf.seek(0)
        tails.append(lf.read()[-4000:])
        lf.close()
    return {'rc': rc, 'logs': tails, 'timed_out': timed_out}


def main():
    for line in sys.stdin:
        line = line.strip()
        if not line:
            continue
        try:
            rep = run(json.loads(line))
        except Exception as e:                   # report, never die silently: the test would hang on readline
            rep = {'rc': [-1], 'logs': [repr(e)], 'timed_out': False}
        sys.stdout.write(json.dumps(rep) + '\n')
        sys.stdout.flush()


if __name__ == '__main__':
    main()
