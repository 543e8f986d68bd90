#!/bin/bash
# End-of-round evidence for one build (full GPU suite with durations, smoke, the contract bench line with cpu_baseline + nominal, the
# profile set named by the sources sha, the fp8 / configs[4] lines): since round 5 an entry of the run table.
exec "$(dirname "$0")/run.sh" r05final "$@"
