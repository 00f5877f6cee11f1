!> ISO_C_BINDING interface to libpolychord_hip.so for Fortran callers.
!!
!! The reference's Fortran programs call the generic `run_polychord` of interfaces_module
!! (src/polychord/interfaces.F90:10-12).  A program that wants the MI355X engine instead binds the same
!! C symbol the reference's C++ facade uses (interfaces.h:2-45 / interfaces.F90:285-383): scalars by value,
!! `comm` by reference, NUL terminated strings.  Callbacks are C-interoperable procedures:
!!   loglikelihood(theta, nDims, phi, nDerived) -> real(c_double)
!!   prior(cube, theta, nDims);  dumper(ndead, nlive, npars, live, dead, logweights, logZ, logZerr)
!! The built-in device likelihoods are selected by passing their exported addresses (polychord_hip_gaussian,
!! ...): the evaluation then runs fused in the sampling kernel.
module polychord_hip
    use iso_c_binding
    implicit none
    private
    public :: polychord_c_interface, polychord_hip_gaussian, polychord_hip_rastrigin, polychord_hip_twin_gaussian, &
              polychord_hip_uniform_prior, polychord_hip_set_gaussian, polychord_hip_set_uniform_prior, &
              polychord_hip_set_option, run_polychord_hip

    interface
        subroutine polychord_c_interface(loglike, prior, dumper, nlive, num_repeats, nprior, nfail, do_clustering, &
                feedback, precision_criterion, logzero, max_ndead, boost_posterior, posteriors, equals, &
                cluster_posteriors, write_resume, write_paramnames, read_resume, write_stats, write_live, write_dead, &
                write_prior, maximise, compression_factor, synchronous, nDims, nDerived, base_dir, file_root, nGrade, &
                grade_frac, grade_dims, n_nlives, loglikes, nlives, seed, comm) bind(c, name="polychord_c_interface")
            import :: c_funptr, c_int, c_bool, c_double, c_char
            type(c_funptr), value :: loglike, prior, dumper
            integer(c_int), value :: nlive, num_repeats, nprior, nfail, feedback, max_ndead, nDims, nDerived, nGrade, &
                                     n_nlives, seed
            logical(c_bool), value :: do_clustering, posteriors, equals, cluster_posteriors, write_resume, &
                                      write_paramnames, read_resume, write_stats, write_live, write_dead, write_prior, &
                                      maximise, synchronous
            real(c_double), value :: precision_criterion, logzero, boost_posterior, compression_factor
            character(kind=c_char) :: base_dir(*), file_root(*)
            real(c_double) :: grade_frac(*), loglikes(*)
            integer(c_int) :: grade_dims(*), nlives(*)
            integer(c_int) :: comm
        end subroutine polychord_c_interface

        function polychord_hip_gaussian(theta, nDims, phi, nDerived) result(logL) bind(c, name="polychord_hip_gaussian")
            import :: c_double, c_int
            real(c_double) :: theta(*), phi(*)
            integer(c_int), value :: nDims, nDerived
            real(c_double) :: logL
        end function
        function polychord_hip_rastrigin(theta, nDims, phi, nDerived) result(logL) bind(c, name="polychord_hip_rastrigin")
            import :: c_double, c_int
            real(c_double) :: theta(*), phi(*)
            integer(c_int), value :: nDims, nDerived
            real(c_double) :: logL
        end function
        function polychord_hip_twin_gaussian(theta, nDims, phi, nDerived) result(logL) &
                bind(c, name="polychord_hip_twin_gaussian")
            import :: c_double, c_int
            real(c_double) :: theta(*), phi(*)
            integer(c_int), value :: nDims, nDerived
            real(c_double) :: logL
        end function
        subroutine polychord_hip_uniform_prior(cube, theta, nDims) bind(c, name="polychord_hip_uniform_prior")
            import :: c_double, c_int
            real(c_double) :: cube(*), theta(*)
            integer(c_int), value :: nDims
        end subroutine
        subroutine polychord_hip_set_gaussian(mu, sigma) bind(c, name="polychord_hip_set_gaussian")
            import :: c_double
            real(c_double), value :: mu, sigma
        end subroutine
        subroutine polychord_hip_set_uniform_prior(nDims, lo, hi) bind(c, name="polychord_hip_set_uniform_prior")
            import :: c_double, c_int
            integer(c_int), value :: nDims
            real(c_double) :: lo(*), hi(*)
        end subroutine
        subroutine polychord_hip_set_option(name, value) bind(c, name="polychord_hip_set_option")
            import :: c_char, c_double
            character(kind=c_char) :: name(*)
            real(c_double), value :: value
        end subroutine
    end interface

contains

    !> Convenience wrapper with the reference's defaults (settings.f90:10-110): files under base_dir/file_root.
    subroutine run_polychord_hip(loglike, prior, nDims, nDerived, nlive, num_repeats, base_dir, file_root, seed, &
                                 do_clustering, write_dead, posteriors)
        type(c_funptr), intent(in) :: loglike, prior
        integer, intent(in) :: nDims, nDerived, nlive, num_repeats, seed
        character(len=*), intent(in) :: base_dir, file_root
        logical, intent(in) :: do_clustering, write_dead, posteriors
        real(c_double) :: grade_frac(1), loglikes(1)
        integer(c_int) :: grade_dims(1), nlives(1), comm
        grade_frac = 1d0; grade_dims = nDims; loglikes = 0d0; nlives = 0; comm = 0
        call polychord_c_interface(loglike, prior, c_null_funptr, int(nlive, c_int), int(num_repeats, c_int), -1_c_int, &
            -1_c_int, logical(do_clustering, c_bool), 0_c_int, 1d-3, -1d30, -1_c_int, 0d0, logical(posteriors, c_bool), &
            .false._c_bool, .false._c_bool, .false._c_bool, .false._c_bool, .false._c_bool, .true._c_bool, .false._c_bool, &
            logical(write_dead, c_bool), .false._c_bool, .false._c_bool, exp(-1d0), .true._c_bool, int(nDims, c_int), &
            int(nDerived, c_int), trim(base_dir)//c_null_char, trim(file_root)//c_null_char, 1_c_int, grade_frac, &
            grade_dims, 0_c_int, loglikes, nlives, int(seed, c_int), comm)
    end subroutine run_polychord_hip

end module polychord_hip
