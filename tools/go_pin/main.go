// go_pin: pins this repository's CPU oracle -- and through it the GPU path -- against the REAL gnark / gnark-crypto at the versions the
// reference pins (gnark_backend_ffi/go.mod:5,23: gnark-crypto v0.9.1, gnark v0.8.0).
//
// NOT COMPILED IN THIS REPOSITORY'S BUILD IMAGE (no Go toolchain, no module cache, no network): this file is the program a maintainer runs
// once on a machine that has them, from the repository root:
//
//	cd tools/go_pin && go mod tidy && go run . ../../tests/golden
//
// It reads the committed fixtures (tests/golden/*.json -- every value in them was produced by oracle/*.py and is what the GPU tests
// compare the device results against) and checks them with upstream code:
//
//	1. fr.Vector wire format            MarshalBinary of the fixture's felts == the fixture's encoding
//	2. MultiExp (G1, G2)                upstream's result on the same points / scalars == the fixture's affine Montgomery image
//	3. fft.Domain FFT / FFTInverse      sha256 of upstream's output memory image == the fixture's, all 8 mode combinations per size
//	4. Groth16                          gnark's VerifyingKey.ReadFrom / Proof.ReadFrom accept the fixture bytes and groth16.Verify accepts the
//	                                    proof; ProvingKey.ReadFrom + WriteTo round-trips the fixture's key image byte for byte
//	5. PLONK                            plonk.VerifyingKey.ReadFrom + InitKZG(kzg.NewSRS(size, alpha)) + plonk.Verify accept the fixture's
//	                                    548-byte proofs (this re-derives gamma, beta, alpha, zeta from the bytes: it pins the proof layout,
//	                                    the transcript and the verifying-key image at once); ProvingKey round trip as for Groth16
//
//	6. HandleValues (f3)                the REFERENCE'S OWN lowering -- plonk_backend.BuildSparseR1CS of gnark_backend_ffi/backend/plonk, which calls
//	                                    backend.HandleValues (backend/common.go:45-76) -- run on the fixtures with two and three public inputs
//	                                    (tests/golden/plonk_multi_public_golden.json), then upstream's plonk.Setup on the fixture's SRS: variable counts,
//	                                    witness order and the ProvingKey / VerifyingKey bytes must be the fixture's layout "reference" (what
//	                                    libgnark_backend.so produces); the proofs of both layouts must verify under their keys.  go.mod points the
//	                                    module gnark_backend_ffi at a checkout of the reference (replace directive: adjust the path).
//
// Every line printed is "PASS ..." or "FAIL ..."; the exit status is the number of failures.  When all pass, DESIGN.md's "parity unpinned"
// can be struck: the oracle is then pinned by upstream itself.
package main

import (
	"bytes"
	"crypto/sha256"
	"encoding/binary"
	"encoding/hex"
	"encoding/json"
	"fmt"
	"math/big"
	"os"
	"path/filepath"

	"github.com/consensys/gnark-crypto/ecc"
	bn254 "github.com/consensys/gnark-crypto/ecc/bn254"
	"github.com/consensys/gnark-crypto/ecc/bn254/fr"
	"github.com/consensys/gnark-crypto/ecc/bn254/fr/fft"
	"github.com/consensys/gnark-crypto/ecc/bn254/fr/kzg"
	"github.com/consensys/gnark/backend/groth16"
	"github.com/consensys/gnark/backend/plonk"
	"github.com/consensys/gnark/backend/witness"

	"gnark_backend_ffi/acir"
	plonk_backend "gnark_backend_ffi/backend/plonk"
)

var failures int

func report(ok bool, format string, args ...interface{}) {
	tag := "PASS"
	if !ok {
		tag = "FAIL"
		failures++
	}
	fmt.Printf("%s %s\n", tag, fmt.Sprintf(format, args...))
}

func must(err error) {
	if err != nil {
		fmt.Println("FATAL", err)
		os.Exit(100)
	}
}

func unhex(s string) []byte {
	b, err := hex.DecodeString(s)
	must(err)
	return b
}

func bigFromHex(s string) *big.Int {
	v, ok := new(big.Int).SetString(s, 16)
	if !ok {
		must(fmt.Errorf("bad hex integer %q", s))
	}
	return v
}

func frFromHex(s string) fr.Element {
	var e fr.Element
	e.SetBigInt(bigFromHex(s))
	return e
}

// SplitMix64 -> 4 little-endian 64-bit limbs -> reduced mod r: the fixtures' PRNG (oracle/bn254_ref.py SplitMix64.felt)
type splitMix struct{ s uint64 }

func (g *splitMix) next() uint64 {
	g.s += 0x9E3779B97F4A7C15
	z := g.s
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9
	z = (z ^ (z >> 27)) * 0x94D049BB133111EB
	return z ^ (z >> 31)
}

func (g *splitMix) felt() fr.Element {
	v := new(big.Int)
	for i := 0; i < 4; i++ {
		limb := new(big.Int).SetUint64(g.next())
		v.Add(v, limb.Lsh(limb, uint(64*i)))
	}
	var e fr.Element
	e.SetBigInt(v) // reduces mod r
	return e
}

func randFelts(seed uint64, n int) []fr.Element {
	g := &splitMix{s: seed}
	out := make([]fr.Element, n)
	for i := range out {
		out[i] = g.felt()
	}
	return out
}

// memory image of Montgomery elements: 4 little-endian uint64 each (what the fixtures hash and what crosses the C ABI)
func limbsLE(words ...uint64) []byte {
	b := make([]byte, 8*len(words))
	for i, w := range words {
		binary.LittleEndian.PutUint64(b[8*i:], w)
	}
	return b
}

func frImage(v []fr.Element) []byte {
	var buf bytes.Buffer
	for i := range v {
		buf.Write(limbsLE(v[i][0], v[i][1], v[i][2], v[i][3]))
	}
	return buf.Bytes()
}

func g1Image(p *bn254.G1Affine) []byte {
	return append(limbsLE(p.X[0], p.X[1], p.X[2], p.X[3]), limbsLE(p.Y[0], p.Y[1], p.Y[2], p.Y[3])...)
}

func g2Image(p *bn254.G2Affine) []byte {
	out := limbsLE(p.X.A0[0], p.X.A0[1], p.X.A0[2], p.X.A0[3])
	out = append(out, limbsLE(p.X.A1[0], p.X.A1[1], p.X.A1[2], p.X.A1[3])...)
	out = append(out, limbsLE(p.Y.A0[0], p.Y.A0[1], p.Y.A0[2], p.Y.A0[3])...)
	return append(out, limbsLE(p.Y.A1[0], p.Y.A1[1], p.Y.A1[2], p.Y.A1[3])...)
}

// ---------------------------------------------------------------------------------------------------------------- fixtures
type msmCase struct {
	Kind         string   `json:"kind"`
	N            int      `json:"n"`
	PointScalars []string `json:"point_scalars"`
	Scalars      []string `json:"scalars"`
	G1           string   `json:"g1"`
	G2           string   `json:"g2"`
}

type nttCase struct {
	Kind       string `json:"kind"`
	LogN       int    `json:"log_n"`
	Seed       uint64 `json:"seed"`
	Inverse    int    `json:"inverse"`
	Decimation int    `json:"decimation"`
	Coset      int    `json:"coset"`
	Sha256     string `json:"sha256"`
}

type groth16Case struct {
	Name    string   `json:"name"`
	NPublic int      `json:"n_public"`
	W       []string `json:"w"`
	Proof   string   `json:"proof"`
}

type bn254Golden struct {
	Msm     []msmCase     `json:"msm"`
	Ntt     []nttCase     `json:"ntt"`
	Groth16 []groth16Case `json:"groth16"`
	Wire    struct {
		Encoded string   `json:"encoded"`
		Felts   []string `json:"felts"`
	} `json:"wire"`
}

type groth16Wire struct {
	Name  string `json:"name"`
	PkHex string `json:"pk_hex"`
	VkHex string `json:"vk_hex"`
}

type plonkCase struct {
	Name     string   `json:"name"`
	NPublic  int      `json:"n_public"`
	Solution []string `json:"solution"`
	SrsAlpha string   `json:"srs_alpha"`
	SrsSize  uint64   `json:"srs_size"`
	Proof    string   `json:"proof"`
	VkHex    string   `json:"vk_hex"`
	PkHex    string   `json:"pk_hex"`
}

func load(path string, into interface{}) {
	raw, err := os.ReadFile(path)
	must(err)
	must(json.Unmarshal(raw, into))
}

// public witness of `values` (no secret part): what groth16.Verify / plonk.Verify take
func publicWitness(values []fr.Element) witness.Witness {
	w, err := witness.New(ecc.BN254.ScalarField())
	must(err)
	ch := make(chan any)
	go func() {
		defer close(ch)
		for _, v := range values {
			ch <- v
		}
	}()
	must(w.Fill(len(values), 0, ch))
	return w
}

// ---------------------------------------------------------------------------------------------------------------- checks
func checkWire(g *bn254Golden) {
	v := make(fr.Vector, len(g.Wire.Felts))
	for i, s := range g.Wire.Felts {
		v[i] = frFromHex(s)
	}
	enc, err := v.MarshalBinary()
	must(err)
	report(hex.EncodeToString(enc) == g.Wire.Encoded, "fr.Vector.MarshalBinary of %d felts", len(v))
	var back fr.Vector
	must(back.UnmarshalBinary(unhex(g.Wire.Encoded)))
	same := len(back) == len(v)
	for i := 0; same && i < len(v); i++ {
		same = back[i].Equal(&v[i])
	}
	report(same, "fr.Vector.UnmarshalBinary round trip")
}

func checkMSM(g *bn254Golden) {
	_, _, g1Gen, g2Gen := bn254.Generators()
	for _, c := range g.Msm {
		p1 := make([]bn254.G1Affine, c.N)
		p2 := make([]bn254.G2Affine, c.N)
		sc := make([]fr.Element, c.N)
		for i := 0; i < c.N; i++ {
			k := bigFromHex(c.PointScalars[i])
			p1[i].ScalarMultiplication(&g1Gen, k) // k = 0 gives the point at infinity, as in the fixture
			p2[i].ScalarMultiplication(&g2Gen, k)
			sc[i] = frFromHex(c.Scalars[i])
		}
		var r1 bn254.G1Affine
		_, err := r1.MultiExp(p1, sc, ecc.MultiExpConfig{})
		must(err)
		report(hex.EncodeToString(g1Image(&r1)) == c.G1, "G1 MultiExp %s n=%d", c.Kind, c.N)
		var r2 bn254.G2Affine
		_, err = r2.MultiExp(p2, sc, ecc.MultiExpConfig{})
		must(err)
		report(hex.EncodeToString(g2Image(&r2)) == c.G2, "G2 MultiExp %s n=%d", c.Kind, c.N)
	}
}

func checkFFT(g *bn254Golden) {
	for _, c := range g.Ntt {
		if c.Kind != "modes" {
			continue
		}
		if c.LogN == 0 {
			continue // a one-point transform is the identity; upstream's kernels are not written for it
		}
		n := 1 << uint(c.LogN)
		a := randFelts(c.Seed, n)
		d := fft.NewDomain(uint64(n))
		dec := fft.DIT // the fixture's `decimation`: 0 = DIT, 1 = DIF (oracle/bn254_ref.py DIT / DIF, same numbering as upstream's iota)
		if c.Decimation == 1 {
			dec = fft.DIF
		}
		if c.Inverse == 1 {
			d.FFTInverse(a, dec, c.Coset == 1)
		} else {
			d.FFT(a, dec, c.Coset == 1)
		}
		sum := sha256.Sum256(frImage(a))
		report(hex.EncodeToString(sum[:]) == c.Sha256, "fft log_n=%d inverse=%d decimation=%d coset=%d", c.LogN, c.Inverse, c.Decimation, c.Coset)
	}
}

func checkGroth16(g *bn254Golden, wires []groth16Wire) {
	byName := map[string]groth16Wire{}
	for _, w := range wires {
		byName[w.Name] = w
	}
	for _, c := range g.Groth16 {
		name := c.Name
		if name == "seq_r1cs_13_r0" { // same key as seq_r1cs_13, proof made with r = s = 0
			name = "seq_r1cs_13"
		}
		w, ok := byName[name]
		if !ok {
			continue
		}
		vk := groth16.NewVerifyingKey(ecc.BN254)
		_, err := vk.ReadFrom(bytes.NewReader(unhex(w.VkHex)))
		report(err == nil, "groth16 VerifyingKey.ReadFrom %s (%v)", c.Name, err)
		if err != nil {
			continue
		}
		proof := groth16.NewProof(ecc.BN254)
		_, err = proof.ReadFrom(bytes.NewReader(unhex(c.Proof)))
		report(err == nil, "groth16 Proof.ReadFrom %s (%v)", c.Name, err)
		if err != nil {
			continue
		}
		pub := make([]fr.Element, 0, c.NPublic)
		for _, s := range c.W[1:c.NPublic] { // wire 0 is the constant ONE, not part of the witness
			pub = append(pub, frFromHex(s))
		}
		err = groth16.Verify(proof, vk, publicWitness(pub))
		report(err == nil, "groth16.Verify accepts the fixture proof %s (%v)", c.Name, err)
	}
	for _, w := range wires {
		pk := groth16.NewProvingKey(ecc.BN254)
		raw := unhex(w.PkHex)
		n, err := pk.ReadFrom(bytes.NewReader(raw))
		report(err == nil && int(n) == len(raw), "groth16 ProvingKey.ReadFrom %s: %d of %d bytes (%v)", w.Name, n, len(raw), err)
		if err != nil {
			continue
		}
		var out bytes.Buffer
		_, err = pk.WriteTo(&out)
		must(err)
		report(bytes.Equal(out.Bytes(), raw), "groth16 ProvingKey.WriteTo(ReadFrom(x)) == x  %s", w.Name)
	}
}

func checkPlonk(cases []plonkCase) {
	for _, c := range cases {
		srs, err := kzg.NewSRS(c.SrsSize, bigFromHex(c.SrsAlpha))
		must(err)
		vk := plonk.NewVerifyingKey(ecc.BN254)
		_, err = vk.ReadFrom(bytes.NewReader(unhex(c.VkHex)))
		report(err == nil, "plonk VerifyingKey.ReadFrom %s (%v)", c.Name, err)
		if err != nil {
			continue
		}
		must(vk.InitKZG(srs))
		proof := plonk.NewProof(ecc.BN254)
		_, err = proof.ReadFrom(bytes.NewReader(unhex(c.Proof)))
		report(err == nil, "plonk Proof.ReadFrom %s (%v)", c.Name, err)
		if err != nil {
			continue
		}
		pub := make([]fr.Element, c.NPublic)
		for i := range pub {
			pub[i] = frFromHex(c.Solution[i])
		}
		err = plonk.Verify(proof, vk, publicWitness(pub))
		report(err == nil, "plonk.Verify accepts the fixture proof %s (%v)", c.Name, err)

		pk := plonk.NewProvingKey(ecc.BN254)
		raw := unhex(c.PkHex)
		n, err := pk.ReadFrom(bytes.NewReader(raw))
		report(err == nil && int(n) == len(raw), "plonk ProvingKey.ReadFrom %s: %d of %d bytes (%v)", c.Name, n, len(raw), err)
		if err != nil {
			continue
		}
		var out bytes.Buffer
		_, err = pk.WriteTo(&out)
		must(err)
		report(bytes.Equal(out.Bytes(), raw), "plonk ProvingKey.WriteTo(ReadFrom(x)) == x  %s", c.Name)

		// kzg.SRS.WriteTo of upstream's SRS starts with G2[0], G2[1]: compare the G1 side with what the fixture's alpha must give
		var sb bytes.Buffer
		_, err = srs.WriteTo(&sb)
		must(err)
		report(sb.Len() == 132+32*int(c.SrsSize), "kzg.SRS.WriteTo length for %d points", c.SrsSize)
	}
}

type multiPublicLayout struct {
	NPublic  int      `json:"n_public"`
	NVars    int      `json:"n_vars"`
	SrsSize  uint64   `json:"srs_size"`
	Solution []string `json:"solution"`
	Proof    string   `json:"proof"`
	VkHex    string   `json:"vk_hex"`
	PkHex    string   `json:"pk_hex"`
}

type multiPublicCase struct {
	Name     string                       `json:"name"`
	Acir     json.RawMessage              `json:"acir"`
	Values   []string                     `json:"values"`
	SrsAlpha string                       `json:"srs_alpha"`
	Layouts  map[string]multiPublicLayout `json:"layouts"`
}

// the reference's own BuildSparseR1CS / HandleValues on circuits with several public inputs, against the fixture's layout "reference"
func checkHandleValues(cases []multiPublicCase) {
	var flat []plonkCase
	for _, c := range cases {
		ref := c.Layouts["reference"]
		var circuit acir.ACIR
		must(json.Unmarshal(c.Acir, &circuit))
		values := make(fr.Vector, len(c.Values))
		for i := range values {
			values[i] = frFromHex(c.Values[i])
		}
		spr, pub, sec := plonk_backend.BuildSparseR1CS(circuit, values)
		report(spr.GetNbPublicVariables() == ref.NPublic && spr.GetNbPublicVariables()+spr.GetNbSecretVariables() == ref.NVars,
			"HandleValues %s: %d public + %d secret variables (fixture: %d of %d)", c.Name, spr.GetNbPublicVariables(), spr.GetNbSecretVariables(), ref.NPublic, ref.NVars)
		got := append(append(fr.Vector{}, pub...), sec...)
		same := len(got) == len(ref.Solution)
		for i := 0; same && i < len(got); i++ {
			w := frFromHex(ref.Solution[i])
			same = w.Equal(&got[i])
		}
		report(same, "HandleValues %s: the witness vector (public, then secret) is the fixture's solution", c.Name)
		srs, err := kzg.NewSRS(ref.SrsSize, bigFromHex(c.SrsAlpha))
		must(err)
		pk, vk, err := plonk.Setup(spr, srs)
		report(err == nil, "plonk.Setup on the reference's constraint system %s (%v)", c.Name, err)
		if err != nil {
			continue
		}
		var pb, vb bytes.Buffer
		_, err = pk.WriteTo(&pb)
		must(err)
		_, err = vk.WriteTo(&vb)
		must(err)
		report(bytes.Equal(vb.Bytes(), unhex(ref.VkHex)), "reference lowering + plonk.Setup: VerifyingKey bytes == fixture (layout reference) %s", c.Name)
		report(bytes.Equal(pb.Bytes(), unhex(ref.PkHex)), "reference lowering + plonk.Setup: ProvingKey bytes == fixture (layout reference) %s", c.Name)
		one := c.Layouts["one_var"]
		report(!bytes.Equal(pb.Bytes(), unhex(one.PkHex)), "the one-variable-per-witness layout is a different key %s", c.Name)
		for lname, l := range c.Layouts {
			flat = append(flat, plonkCase{Name: c.Name + "/" + lname, NPublic: l.NPublic, Solution: l.Solution, SrsAlpha: c.SrsAlpha, SrsSize: l.SrsSize, Proof: l.Proof,
				VkHex: l.VkHex, PkHex: l.PkHex})
		}
	}
	checkPlonk(flat)
}

func main() {
	dir := "../../tests/golden"
	if len(os.Args) > 1 {
		dir = os.Args[1]
	}
	var g bn254Golden
	load(filepath.Join(dir, "bn254_golden.json"), &g)
	var gw []groth16Wire
	load(filepath.Join(dir, "groth16_wire_golden.json"), &gw)
	var pc []plonkCase
	load(filepath.Join(dir, "plonk_golden.json"), &pc)

	checkWire(&g)
	checkMSM(&g)
	checkFFT(&g)
	checkGroth16(&g, gw)
	checkPlonk(pc)
	var mp []multiPublicCase
	load(filepath.Join(dir, "plonk_multi_public_golden.json"), &mp)
	checkHandleValues(mp)
	fmt.Printf("%d failure(s)\n", failures)
	os.Exit(failures)
}
