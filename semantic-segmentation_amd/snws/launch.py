"""Start one process per GPU from a plain `python script.py --gpus N` invocation.

The reference needs no launcher for its multi-GPU mode (nn.DataParallel inside one process, train.py:210-214); the
one-process-per-GPU design does, so a script that is started without a launcher starts its own ranks: the parent -
which never touches the GPU - runs `python -m torch.distributed.run` as a CHILD process (no exec of a process that
has initialised the GPU), relays the single result line rank 0 prints on stdout, and exits with the child's code.
"""
import os
import socket
import subprocess
import sys


def launched():
    """True inside a rank started by torch.distributed.run (or any launcher that sets WORLD_SIZE)."""
    return "WORLD_SIZE" in os.environ


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(script, argv, nproc, out=None, err=None, env=None):
    """Run `script argv...` as nproc ranks on this node.  Lines of the ranks' stdout that look like a JSON object
    are written to `out` (default sys.stdout); everything else they print goes to `err` (default sys.stderr).
    Returns the launcher's exit code (non-zero when any rank failed)."""
    out = out or sys.stdout
    err = err or sys.stderr
    e = dict(os.environ if env is None else env)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    e.setdefault("OMP_NUM_THREADS", str(max(1, min(16, (os.cpu_count() or 8) // max(1, nproc)))))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(int(nproc)),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), script] + list(argv)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=None if err is sys.stderr else subprocess.STDOUT,
                            text=True, env=e)
    try:
        for line in proc.stdout:
            s = line.strip()
            if s.startswith("{") and s.endswith("}"):
                out.write(s + "\n")
                out.flush()
            else:
                err.write(line)
                err.flush()
        return proc.wait()
    except BaseException:
        proc.terminate()
        try:
            proc.wait(timeout=30)
        except subprocess.TimeoutExpired:
            proc.kill()
        raise
