import itertools, numpy as np
groups = [list(range(0,4))+list(range(12,16))+list(range(20,28)),
          list(range(4,12))+list(range(16,20))+list(range(28,32))]
groups += [[l+32 for l in g] for g in groups[:2]]
def conflicts(f, shifts=(0,1,2), bases=range(0,32,16), kks=(0,1)):
    worst = 0
    for sh in shifts:
        for base in bases:
            for kk in kks:
                for g in groups:
                    banks = {}
                    for l in g:
                        frow, fq = l & 15, l >> 4
                        row = base + frow + sh
                        ch = (fq ^ f(row)) ^ (kk << 2)
                        addr = row * 128 + ch * 16
                        b = (addr // 16) % 16   # 16-B slot within 256-B bank row
                        banks[b] = banks.get(b, 0) + 1
                    worst = max(worst, max(banks.values()))
    return worst
print("current", conflicts(lambda r: (r >> 1) & 7))
# linear family: f bits = XOR of selected row bits (rows mod 64 -> 6 bits)
best = []
for m in itertools.product(range(64), repeat=3):
    def f(r, m=m):
        v = 0
        for b in range(3):
            v |= (bin(r & m[b]).count("1") & 1) << b
        return v
    c = conflicts(f, bases=range(0, 64, 16))
    if c == 1:
        best.append(m)
print(len(best), best[:20])
