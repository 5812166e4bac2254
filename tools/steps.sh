#!/bin/bash
# Run the given commands one after another on the GPU box; an ordinary failure (a failing test)
# does not stop the sequence, a step that timed out or was killed does (no further GPU work after
# a hang).  usage: tools/steps.sh 'cmd1' 'cmd2' ...
rc_all=0
for c in "$@"; do
  echo "== $c"
  bash -o pipefail -c "$c"
  rc=$?
  echo "== rc $rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -eq 143 ]; then echo "== step timed out / was killed: stopping"; exit $rc; fi
  [ $rc -ne 0 ] && rc_all=$rc
done
exit $rc_all
