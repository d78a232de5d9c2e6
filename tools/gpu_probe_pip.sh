#!/bin/bash
# Round 5, VERDICT "Next" #1 step A: does the GPU box have ANY route to mujoco / mujoco-mjx / jax?
# Time-boxed; writes gpurun_out/r5probe/probe.txt.  Nothing here installs system-wide.
out=gpurun_out/r5probe; mkdir -p $out
{
echo "== date"; date -u
echo "== pip config"; python3 -m pip config list 2>&1 | head
echo "== pip download (20 s)"; timeout 25 python3 -m pip download mujoco mujoco-mjx "jax[cpu]" -d /tmp/w --no-deps --timeout 5 --retries 0 2>&1 | tail -5
echo "== rc $?"
echo "== pip index"; timeout 15 python3 -m pip index versions mujoco --timeout 5 --retries 0 2>&1 | tail -3
echo "== DNS / route"; timeout 5 getent hosts pypi.org 2>&1; timeout 5 python3 -c "import socket; s=socket.create_connection(('pypi.org',443),3); print('tcp ok')" 2>&1 | tail -1
echo "== env proxies"; env | grep -i -E 'proxy|pip_|index' 
echo "== local wheels"; find / -xdev \( -iname '*mujoco*' -o -iname 'jax-*' -o -iname 'jaxlib*' -o -iname '*brax*' -o -iname '*.whl' \) -not -path '/proc/*' 2>/dev/null | head -20
echo "== importable"; for m in mujoco jax jaxlib brax mujoco_playground onnxruntime; do python3 -c "import $m; print('$m', getattr($m,'__version__','?'))" 2>&1 | tail -1; done
echo "== nproc / mem"; nproc; free -g | head -2
} > $out/probe.txt 2>&1
cat $out/probe.txt
