"""Helpers of bench.py that are not the contract line: side workloads, the one-GPU child process, post-mortem of GPU faults."""
