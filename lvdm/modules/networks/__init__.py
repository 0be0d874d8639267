"""Import-path shim: the reference's yaml configs name classes by these dotted paths."""
