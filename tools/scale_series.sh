#!/bin/bash
# north_star's 1 / 2 / 4 / 8 GPU table (absolute, fraction of roofline, reference CPU path beside it): see tools/scale_series.py
cd "$(dirname "$0")/.." && exec python tools/scale_series.py "$@"
