"""Measurement helpers (see tools/README.md); a package only so the scripts can share `replay_time`."""
