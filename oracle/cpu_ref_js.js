"use strict";
// cpu_ref_js.js -- TEST / MEASUREMENT INFRASTRUCTURE (oracle/, never on the product path).
//
// Single-thread JavaScript restatement of what snarkjs@0.1.20 `groth.genProof` computes (SURVEY.md App. B;
// snarkjs is pinned at /root/reference/operator/yarn.lock:5674-5675 but is not vendored and cannot be
// installed here, so this is a restatement of its published algorithm, NOT snarkjs itself):
//   * one scalar multiplication per signal and per query (double-and-add over Jacobian coordinates, the shape
//     of snarkjs' GCurve.mulScalar) -- no Pippenger, no windowing;
//   * calculateH through the coefficient product of A and B (size-2m FFTs), upper half;
//   * native BigInt arithmetic (snarkjs 0.1.20 uses big-integer, which delegates to native BigInt when present).
// Input : JSON {pk: snarkjs-shaped proving key (decimal strings), witness: [...], r, s}
// Output: JSON {proof: {pi_a, pi_b, pi_c}, seconds: {h, msm, total}}
// It exists to put a "snarkjs-equivalent single thread" time next to the GPU number (bench.py cpu_baseline) and
// is pinned by tests/test_oracle.py against the toxic-waste closed form.
const fs = require("fs");

const Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583n;
const R = 21888242871839275222246405745257275088548364400416034343698204186575808495617n;

function modInv(a, p) {
  let [t, nt, r, nr] = [0n, 1n, p, ((a % p) + p) % p];
  while (nr !== 0n) { const q = r / nr; [t, nt] = [nt, t - q * nt]; [r, nr] = [nr, r - q * nr]; }
  return ((t % p) + p) % p;
}
function modPow(b, e, p) { let r = 1n; b %= p; while (e > 0n) { if (e & 1n) r = (r * b) % p; b = (b * b) % p; e >>= 1n; } return r; }

// ---- fields: F1 = Fq, F2 = Fq[u]/(u^2+1)
const F1 = {
  zero: 0n, one: 1n,
  add: (a, b) => (a + b) % Q, sub: (a, b) => (a - b + Q) % Q, mul: (a, b) => (a * b) % Q,
  square: (a) => (a * a) % Q, neg: (a) => (a === 0n ? 0n : Q - a), inv: (a) => modInv(a, Q),
  isZero: (a) => a === 0n, eq: (a, b) => a === b, fromJson: (v) => BigInt(v), toJson: (v) => v.toString(),
};
const F2 = {
  zero: [0n, 0n], one: [1n, 0n],
  add: (a, b) => [(a[0] + b[0]) % Q, (a[1] + b[1]) % Q],
  sub: (a, b) => [(a[0] - b[0] + Q) % Q, (a[1] - b[1] + Q) % Q],
  mul: (a, b) => [(((a[0] * b[0] - a[1] * b[1]) % Q) + Q) % Q, (a[0] * b[1] + a[1] * b[0]) % Q],
  square: (a) => [(((a[0] * a[0] - a[1] * a[1]) % Q) + Q) % Q, (2n * a[0] * a[1]) % Q],
  neg: (a) => [a[0] === 0n ? 0n : Q - a[0], a[1] === 0n ? 0n : Q - a[1]],
  inv: (a) => { const n = modInv((a[0] * a[0] + a[1] * a[1]) % Q, Q); return [(a[0] * n) % Q, (a[1] === 0n ? 0n : ((Q - a[1]) * n) % Q)]; },
  isZero: (a) => a[0] === 0n && a[1] === 0n, eq: (a, b) => a[0] === b[0] && a[1] === b[1],
  fromJson: (v) => [BigInt(v[0]), BigInt(v[1])], toJson: (v) => [v[0].toString(), v[1].toString()],
};

// ---- short Weierstrass a = 0, Jacobian (x, y, z); z = 0 is the point at infinity (snarkjs: [0, 1, 0])
function curve(F) {
  const zero = [F.zero, F.one, F.zero];
  const isZero = (p) => F.isZero(p[2]);
  function double(p) {
    if (isZero(p)) return p;
    const A = F.square(p[0]), B = F.square(p[1]), C = F.square(B);
    let D = F.sub(F.sub(F.square(F.add(p[0], B)), A), C);
    D = F.add(D, D);
    const E = F.add(F.add(A, A), A), Fq_ = F.square(E);
    const x3 = F.sub(Fq_, F.add(D, D));
    let C8 = F.add(C, C); C8 = F.add(C8, C8); C8 = F.add(C8, C8);
    const y3 = F.sub(F.mul(E, F.sub(D, x3)), C8);
    const yz = F.mul(p[1], p[2]);
    return [x3, y3, F.add(yz, yz)];
  }
  function add(p, q) {
    if (isZero(p)) return q;
    if (isZero(q)) return p;
    const z1z1 = F.square(p[2]), z2z2 = F.square(q[2]);
    const u1 = F.mul(p[0], z2z2), u2 = F.mul(q[0], z1z1);
    const s1 = F.mul(F.mul(p[1], q[2]), z2z2), s2 = F.mul(F.mul(q[1], p[2]), z1z1);
    if (F.eq(u1, u2)) return F.eq(s1, s2) ? double(p) : zero;
    const H = F.sub(u2, u1), I = F.square(F.add(H, H)), J = F.mul(H, I);
    let r = F.sub(s2, s1); r = F.add(r, r);
    const V = F.mul(u1, I);
    const x3 = F.sub(F.sub(F.square(r), J), F.add(V, V));
    const s1J = F.mul(s1, J);
    const y3 = F.sub(F.mul(r, F.sub(V, x3)), F.add(s1J, s1J));
    const z3 = F.mul(F.sub(F.sub(F.square(F.add(p[2], q[2])), z1z1), z2z2), H);
    return [x3, y3, z3];
  }
  function mulScalar(p, e) {  // plain double-and-add, most significant bit first
    let res = zero;
    const bits = e.toString(2);
    for (let i = 0; i < bits.length; i++) {
      res = double(res);
      if (bits[i] === "1") res = add(res, p);
    }
    return res;
  }
  function affine(p) {
    if (isZero(p)) return null;
    const zi = F.inv(p[2]), zi2 = F.square(zi);
    return [F.mul(p[0], zi2), F.mul(p[1], F.mul(zi2, zi))];
  }
  const fromJson = (v) => [F.fromJson(v[0]), F.fromJson(v[1]), F.fromJson(v[2])];
  return { zero, isZero, double, add, mulScalar, affine, fromJson };
}
const G1 = curve(F1), G2 = curve(F2);

// ---- Fr FFT (radix 2, recursive decimation in time is what snarkjs' PolField does; iterative here, same result)
function rootOfUnity(n) { return modPow(5n, (R - 1n) / BigInt(n), R); }
function fft(a, invert) {
  const n = a.length;
  const out = a.slice();
  for (let i = 1, j = 0; i < n; i++) {
    let bit = n >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) { const t = out[i]; out[i] = out[j]; out[j] = t; }
  }
  for (let len = 2; len <= n; len <<= 1) {
    let w = rootOfUnity(len);
    if (invert) w = modInv(w, R);
    for (let i = 0; i < n; i += len) {
      let wn = 1n;
      for (let k = 0; k < len / 2; k++) {
        const u = out[i + k], v = (out[i + k + len / 2] * wn) % R;
        out[i + k] = (u + v) % R;
        out[i + k + len / 2] = (u - v + R) % R;
        wn = (wn * w) % R;
      }
    }
  }
  if (invert) { const ni = modInv(BigInt(n), R); for (let i = 0; i < n; i++) out[i] = (out[i] * ni) % R; }
  return out;
}

function calculateH(pk, w) {
  const m = pk.domainSize;
  const a = new Array(m).fill(0n), b = new Array(m).fill(0n);
  for (let s = 0; s < pk.nVars; s++) {
    if (w[s] === 0n) continue;
    for (const c of Object.keys(pk.polsA[s])) a[c] = (a[c] + BigInt(pk.polsA[s][c]) * w[s]) % R;
    for (const c of Object.keys(pk.polsB[s])) b[c] = (b[c] + BigInt(pk.polsB[s][c]) * w[s]) % R;
  }
  const pad = (v) => v.concat(new Array(m).fill(0n));
  const ea = fft(pad(fft(a, true)), false), eb = fft(pad(fft(b, true)), false);
  const prod = fft(ea.map((x, i) => (x * eb[i]) % R), true);
  return prod.slice(m);  // (A*B - C) / Z for a satisfying witness: C has degree < m and does not reach this half
}

function genProof(pk, witnessIn, r, s) {
  const t0 = Date.now();
  const w = witnessIn.map((x) => ((BigInt(x) % R) + R) % R);
  const h = calculateH(pk, w);
  const t1 = Date.now();
  let piA = G1.zero, piB = G2.zero, piB1 = G1.zero, piC = G1.zero;
  for (let sI = 0; sI < pk.nVars; sI++) {
    if (w[sI] === 0n) continue;
    piA = G1.add(piA, G1.mulScalar(G1.fromJson(pk.A[sI]), w[sI]));
    piB1 = G1.add(piB1, G1.mulScalar(G1.fromJson(pk.B1[sI]), w[sI]));
    piB = G2.add(piB, G2.mulScalar(G2.fromJson(pk.B2[sI]), w[sI]));
    if (sI > pk.nPublic) piC = G1.add(piC, G1.mulScalar(G1.fromJson(pk.C[sI]), w[sI]));
  }
  for (let i = 0; i < h.length; i++) if (h[i] !== 0n) piC = G1.add(piC, G1.mulScalar(G1.fromJson(pk.hExps[i]), h[i]));
  const delta1 = G1.fromJson(pk.vk_delta_1), delta2 = G2.fromJson(pk.vk_delta_2);
  piA = G1.add(G1.add(piA, G1.fromJson(pk.vk_alfa_1)), G1.mulScalar(delta1, r));
  piB = G2.add(G2.add(piB, G2.fromJson(pk.vk_beta_2)), G2.mulScalar(delta2, s));
  piB1 = G1.add(G1.add(piB1, G1.fromJson(pk.vk_beta_1)), G1.mulScalar(delta1, s));
  piC = G1.add(piC, G1.mulScalar(piA, s));
  piC = G1.add(piC, G1.mulScalar(piB1, r));
  piC = G1.add(piC, G1.mulScalar(delta1, (R - (r * s) % R) % R));
  const t2 = Date.now();
  const a = G1.affine(piA), b = G2.affine(piB), c = G1.affine(piC);
  return {
    proof: { pi_a: [a[0].toString(), a[1].toString(), "1"], pi_b: [F2.toJson(b[0]), F2.toJson(b[1]), ["1", "0"]], pi_c: [c[0].toString(), c[1].toString(), "1"] },
    seconds: { h: (t1 - t0) / 1e3, msm: (t2 - t1) / 1e3, total: (t2 - t0) / 1e3 },
  };
}

if (require.main === module) {
  const d = JSON.parse(fs.readFileSync(process.argv[2]));
  const out = genProof(d.pk, d.witness, BigInt(d.r), BigInt(d.s));
  process.stdout.write(JSON.stringify(out) + "\n");
}
module.exports = { genProof };
