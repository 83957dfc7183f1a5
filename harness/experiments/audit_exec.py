"""Round 6: who starts a program from inside bench.py?  Runs a script in-process with a Python audit hook that
prints the stack of every spawn / exec / fork (subprocess.Popen, os.exec*, os.posix_spawn, os.fork, os.system).
    python harness/experiments/audit_exec.py bench.py --no-cpu-baseline --no-reference-formats"""
import runpy
import sys
import traceback

EVENTS = ("subprocess.Popen", "os.exec", "os.posix_spawn", "os.fork", "os.forkpty", "os.system", "os.spawn", "os.startfile")


def hook(event, args):
    if event in EVENTS:
        sys.stderr.write(f"AUDIT {event} {args[:2]!r}\n")
        traceback.print_stack(limit=14, file=sys.stderr)


sys.addaudithook(hook)
script = sys.argv[1]
sys.argv = sys.argv[1:]
runpy.run_path(script, run_name="__main__")
