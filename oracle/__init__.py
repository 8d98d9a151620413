"""Test infrastructure, never the product: CPU restatements of the reference's hot path (np_oracle: NumPy fp64;
torch_port: the reference's op sequence in PyTorch CPU ops).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg import this package; cl_ica_amd never does (tests/test_host_logic.py checks it)."""
