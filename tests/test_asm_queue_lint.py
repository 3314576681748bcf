"""CPU: the generated gfx950 code of k_logmel_h and k_logmel_h_clip (one body, two instantiations) keeps to the rules its hand-counted load queue depends on (tools/check_asm_queue.py:
no compiler-generated instruction touches the queue's fixed registers once the queue runs; every asm block waits before it reads
a slot and refills it afterwards; slots in cyclic order).  Compiles csrc/embed.hip to assembly (hipcc cross-compiles without a GPU)."""
import importlib.util
import os
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc") and shutil.which("hipcc") is None, reason="needs hipcc")
def test_logmel_queue_registers_are_left_alone(capsys):
    spec = importlib.util.spec_from_file_location("check_asm_queue", os.path.join(ROOT, "tools", "check_asm_queue.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    import sys
    argv, sys.argv = sys.argv, ["check_asm_queue.py"]
    try:
        rc = mod.main()
    finally:
        sys.argv = argv
    out = capsys.readouterr().out
    assert rc == 0, out
    assert out.count("slot order 012340123401234") == 2, out          # both kernels
