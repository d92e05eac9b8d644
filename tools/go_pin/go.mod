module zkmi/go_pin

go 1.18

// the versions the reference pins (gnark_backend_ffi/go.mod:5,23); `go mod tidy` fills in the indirect requirements and go.sum
require (
	github.com/consensys/gnark v0.8.0
	github.com/consensys/gnark-crypto v0.9.1
	gnark_backend_ffi v0.0.0
)

// check 6 runs the reference's own lowering: point this at a checkout of lambdaclass/noir_backend_using_gnark (its Go module is gnark_backend_ffi/)
replace gnark_backend_ffi => ../../../reference/gnark_backend_ffi
