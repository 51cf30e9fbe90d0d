"""Measurement infrastructure of bench.py (NOT the product: roargraph_amd/ never imports it, and only here may the CPU checker under oracle/ be
timed): workloads and legs of the default run, the one-GPU child process, post-mortem of GPU faults, the lifecycle stress."""
