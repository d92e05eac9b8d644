module zkmi/go_pin

go 1.18

// the versions the reference pins (gnark_backend_ffi/go.mod:5,23); `go mod tidy` fills in the indirect requirements and go.sum
require (
	github.com/consensys/gnark v0.8.0
	github.com/consensys/gnark-crypto v0.9.1
)
