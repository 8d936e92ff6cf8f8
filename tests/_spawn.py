"""Test infrastructure: child processes (bench.py, torch.distributed.run) whose failures must be READABLE in a record taken on
a box nobody can log into.  `report` keeps the head AND the tail of both streams and pulls the ranks' own tracebacks (tagged by
bench.py's __main__) to the front; `run_ranks` launches bench.py under torch.distributed.run the way the driver does, pinned to
the loopback, with one more attempt on a fresh port when a launch failed."""
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clip(text, head=1500, tail=3500):
    if len(text) <= head + tail:
        return text
    return text[:head] + f"\n   [... {len(text) - head - tail} characters cut ...]\n" + text[-tail:]


def report(proc, what=""):
    """everything a reader of the assertion message needs: command, return code, the ranks' own tracebacks, both streams"""
    cmd = proc.args if isinstance(proc.args, str) else " ".join(map(str, proc.args))
    seen, cause = set(), []
    for line in (proc.stdout or "").splitlines() + (proc.stderr or "").splitlines():
        if "bench.py[FAILED]" in line and line not in seen:
            seen.add(line)
            cause.append(line)
    return (f"{what} rc={proc.returncode}\n$ {cmd}\n--- cause (rank tracebacks) ---\n" + "\n".join(cause[-60:]) +
            f"\n--- stdout ---\n{_clip(proc.stdout or '')}\n--- stderr ---\n{_clip(proc.stderr or '')}")


def free_port():
    with socket.socket() as s:
        s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def loopback_env(**extra):
    """one node: every rendezvous and every gloo pair over 127.0.0.1, whatever the box's hostname resolves to"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", GLOO_SOCKET_IFNAME="lo")
    env.update(extra)
    return env


def run_bench(args, timeout=600, env=None):
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), capture_output=True, text=True, timeout=timeout, cwd=ROOT,
                          env=env or dict(os.environ))


def run_ranks(nproc, bench_args, timeout=600, attempts=2, **env_extra):
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...` (the
    driver's command line).  A failed launch gets ONE more attempt on another port (rendezvous trouble, a port taken between
    probing and binding); every failed attempt's full report is kept for the assertion message.
    Returns (CompletedProcess of the last attempt, [reports of the failed attempts])."""
    failed = []
    out = None
    for attempt in range(attempts):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), os.path.join(ROOT, "bench.py")] + list(bench_args)
        out = subprocess.run(cmd, env=loopback_env(**env_extra), cwd=ROOT, capture_output=True, text=True, timeout=timeout)
        if out.returncode == 0:
            break
        failed.append(report(out, f"attempt {attempt + 1}"))
    return out, failed
