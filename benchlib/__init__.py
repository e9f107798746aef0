"""Parts of bench.py (repo root): launcher, runner, roofline accounting, side measurements, CPU baseline / parity block.
bench.py keeps the contract (flags, the one JSON line); nothing here is imported by stmask_amd/."""
