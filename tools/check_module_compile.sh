#!/bin/bash
# Diagnostic for INTEGRATION.md section 2 (build container only: it reads /root/reference).
# Syntax-checks the reference's own callers of the hot path, src/Module/*.cpp, against THIS repo's include/ygz headers and
# lists what does not resolve.  boost/format.hpp and boost/timer.hpp are absent from the image; two EMPTY files of those
# names are created in a temporary directory for the duration of the check so that the compiler gets past the #include
# lines (nothing is built, linked or kept -- g++ -fsyntax-only).
# Expected result (recorded in INTEGRATION.md): every hot-path call and (round 6) the covisibility members of Frame resolve; what remains
# is out of scope by SURVEY 2.1 -- the Initializer, highgui drawing (cv::circle, putText, imshow, waitKey, Scalar, CV_FONT_*) and boost::format.
set -u
REF=${1:-/root/reference}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
[ -d "$REF/src/Module" ] || { echo "no reference tree at $REF (this check only runs in the build container)"; exit 0; }
TMP=$(mktemp -d); trap 'rm -rf "$TMP"' EXIT
mkdir -p "$TMP/boost"; : > "$TMP/boost/format.hpp"; : > "$TMP/boost/timer.hpp"
rc=0
for f in "$REF"/src/Module/*.cpp; do
    echo "== ${f#$REF/}"
    g++ -std=c++17 -fsyntax-only -I "$ROOT/include" -I "$TMP" -I "$REF/include" "$f" 2>&1 | grep -E "error" | sed -e "s#$REF/##" -e "s#$ROOT/##" > "$TMP/err.txt"
    total=$(wc -l < "$TMP/err.txt")
    oos=$(grep -cE "Initializer|_init|circle|Scalar|putText|imshow|waitKey|CV_FONT|boost|fmt|<type error>" "$TMP/err.txt")
    echo "   $total unresolved, $oos of them out of scope (Initializer / highgui / boost::format)"
    grep -vE "Initializer|_init|circle|Scalar|putText|imshow|waitKey|CV_FONT|boost|fmt|<type error>" "$TMP/err.txt" | sed 's/^/   IN SCOPE: /'
    [ "$total" -eq "$oos" ] || rc=1
done
exit $rc
