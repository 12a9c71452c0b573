python tools/sort_potential.py C5 2>/dev/null | tail -1
python tools/sort_potential.py C3 2>/dev/null | tail -1
