"""ZPAQL configurations used by the context-mixing parity tests (syntax: ZSFX/libzpaq.h:685-751).
Together they touch every component type."""

ORDER1_CM = """comp 0 0 0 0 1
  0 cm 16 255
hcomp
  *d=a halt
end
"""

# the "mid" configuration from the ZPAQ documentation: icm, isse chain, match, mix
MID = """comp 3 3 0 0 8
  0 icm 5
  1 isse 13 0
  2 isse 17 1
  3 isse 18 2
  4 isse 18 3
  5 isse 19 4
  6 match 22 24
  7 mix 16 0 7 24 255
hcomp
  c++ *c=a b=c a=0
  d= 1 hash *d=a
  b-- d++ hash *d=a
  b-- d++ hash *d=a
  b-- d++ hash *d=a
  b-- d++ hash *d=a
  b-- d++ hash b-- hash *d=a
  d++ a=*c a<<= 8 *d=a
  halt
end
"""

# every remaining type: cons, cm, avg, mix2, sse (+ icm), with arithmetic in HCOMP
ALLTYPES = """comp 3 8 0 0 8
  0 const 140
  1 cm 18 20
  2 icm 12
  3 avg 1 2 100
  4 mix2 10 1 2 30 255
  5 sse 12 4 8 200
  6 cm 9 255
  7 mix2 0 5 6 16 0
hcomp
  c++ *c=a b=c
  d= 1 a=*b hash *d=a
  d++ b-- hash *d=a
  d= 4 a=*c a*= 7 a+= 3 a&= 255 *d=a
  d++ a=*c a<<= 3 a^=*b a%= 251 *d=a
  d++ a=*c a>>= 2 a|= 64 a/= 3 *d=a
  d++ a=*c a== 32 if a= 9 else a= 1 endif *d=a
  d++ *d=0
  halt
end
"""

ALL = {"order1_cm": ORDER1_CM, "mid": MID, "alltypes": ALLTYPES}
