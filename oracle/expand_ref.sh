#!/bin/sh
# Test infrastructure.  Expands a *.cpp.in template on stdout: a line
#     //@REF <file> <first> <last> <anchor text>
# is replaced by lines <first>..<last> of $REF/<file>, read where the file lies in the reference checkout; every other
# line is copied.  <anchor text> (rest of the line) must occur in line <first> of that file -- a guard against a
# reference checkout whose line numbers differ from the ones the template was written for.  The expansion is piped
# straight into the compiler (oracle/Makefile): the reference's text is never written into the repository or into
# oracle/_ref/, only the compiled .so is.
#
# usage: expand_ref.sh <reference root> <template>
REF="$1"
TPL="$2"
[ -f "$TPL" ] || { echo "expand_ref.sh: no template $TPL" >&2; exit 2; }
fail=0
n=0
while IFS= read -r line; do
  n=$((n + 1))
  case "$line" in
    "//@REF "*)
      set -f
      # shellcheck disable=SC2086
      set -- $line
      set +f
      file="$2"; a="$3"; b="$4"
      shift 4
      anchor="$*"
      [ -f "$REF/$file" ] || { echo "expand_ref.sh: $REF/$file missing" >&2; exit 2; }
      if [ -n "$anchor" ] && ! sed -n "${a}p" "$REF/$file" | grep -qF -- "$anchor"; then
        echo "expand_ref.sh: $file:$a does not contain '$anchor' (template written for another checkout)" >&2
        fail=1
      fi
      printf '#line %s "%s"\n' "$a" "$REF/$file"
      sed -n "${a},${b}p" "$REF/$file"
      printf '#line %s "%s"\n' "$((n + 1))" "$TPL"
      ;;
    *)
      printf '%s\n' "$line"
      ;;
  esac
done < "$TPL"
exit $fail
