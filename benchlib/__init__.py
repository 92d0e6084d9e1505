"""bench.py's parts (bench.py at the repository root keeps the driver contract and main()).  Test / measurement infrastructure:
the package acados_amd never imports it."""
