!> Fortran driver in the style of src/drivers/polychord_examples.f90: the 20-D Gaussian of ini/gaussian.ini on the
!! MI355X engine, once with the device likelihood and once with a likelihood written in Fortran (host callback).
program example_gaussian
    use iso_c_binding
    use polychord_hip
    implicit none
    integer, parameter :: nDims = 4, nDerived = 1
    call polychord_hip_set_gaussian(0.5d0, 0.1d0)
    call run_polychord_hip(c_funloc(polychord_hip_gaussian), c_funloc(polychord_hip_uniform_prior), nDims, nDerived, 200, 8, &
                           "chains", "f_device", 3, .false., .true., .true.)
    call run_polychord_hip(c_funloc(my_loglike), c_funloc(my_prior), nDims, nDerived, 100, 8, "chains", "f_host", 3, &
                           .false., .true., .false.)
contains
    function my_loglike(theta, n, phi, nd) result(logL) bind(c)
        integer(c_int), value :: n, nd
        real(c_double) :: theta(n), phi(nd)
        real(c_double) :: logL
        logL = -n * (log(0.1d0) + 0.5d0 * log(8d0 * atan(1d0))) - sum((theta - 0.5d0)**2) / (2d0 * 0.1d0**2)
        if (nd >= 1) phi(1) = sqrt(sum((theta - 0.5d0)**2))
    end function
    subroutine my_prior(cube, theta, n) bind(c)
        integer(c_int), value :: n
        real(c_double) :: cube(n), theta(n)
        theta = cube
    end subroutine
end program example_gaussian
