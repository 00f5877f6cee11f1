! ref_units.f90 -- scratch harness (our code) that calls the REFERENCE's own Fortran modules
! (compiled by oracle/Makefile from /root/reference/src/polychord) on fixed inputs and prints the
! results as JSON.  Used once, in the build container, by oracle/gen_golden.py to produce
! tests/golden/ref_units.json (numbers only).  Test infrastructure.
program ref_units
    use utils_module, only: dp, calc_cholesky, logsumexp, logaddexp, inv_normal_cdf, relabel
    use KNN_clustering, only: NN_clustering, compute_knn
    use calculate_module, only: calculate_similarity_matrix
    implicit none
    real(dp) :: a(3,3), L(3,3), v(3), p(5), x(2,12), S(12,12), b4(4,4), L4(4,4)
    integer :: labels(12), ncl, i, j, knn(4,12), lab2(8), nl
    a = reshape([4d0,2d0,.6d0, 2d0,2d0,.5d0, .6d0,.5d0,3d0],[3,3])
    L = calc_cholesky(a)
    write(*,'(A)') '{'
    write(*,'(A)',advance='no') '"cholesky3": ['
    do i=1,3
        do j=1,3
            write(*,'(ES24.16)',advance='no') L(i,j)
            if (.not.(i==3.and.j==3)) write(*,'(A)',advance='no') ','
        end do
    end do
    write(*,'(A)') '],'
    ! not positive definite -> identity * sqrt(trace)
    b4 = 0d0
    do i=1,4
        b4(i,i) = 1d0
    end do
    b4(1,2) = 2d0; b4(2,1) = 2d0
    L4 = calc_cholesky(b4)
    write(*,'(A)',advance='no') '"cholesky_fallback4": ['
    do i=1,4
        do j=1,4
            write(*,'(ES24.16)',advance='no') L4(i,j)
            if (.not.(i==4.and.j==4)) write(*,'(A)',advance='no') ','
        end do
    end do
    write(*,'(A)') '],'
    p = [0.001d0, 0.3d0, 0.975d0, 1d-12, 0.5d0]
    v = 0
    write(*,'(A)',advance='no') '"as241_p": ['
    do i=1,5
        write(*,'(ES24.16)',advance='no') p(i)
        if (i<5) write(*,'(A)',advance='no') ','
    end do
    write(*,'(A)') '],'
    p = inv_normal_cdf(p)
    write(*,'(A)',advance='no') '"as241_x": ['
    do i=1,5
        write(*,'(ES24.16)',advance='no') p(i)
        if (i<5) write(*,'(A)',advance='no') ','
    end do
    write(*,'(A)') '],'
    write(*,'(A,ES24.16,A)') '"logsumexp_m1000": ', logsumexp([-1000d0,-1001d0,-1002d0]), ','
    write(*,'(A,ES24.16,A)') '"logaddexp_3_5": ', logaddexp(3d0,5d0), ','
    ! three well separated blobs of 4 points in 2-D
    x(:,1)=[0.10d0,0.10d0]; x(:,2)=[0.12d0,0.11d0]; x(:,3)=[0.09d0,0.13d0]; x(:,4)=[0.11d0,0.08d0]
    x(:,5)=[0.80d0,0.82d0]; x(:,6)=[0.81d0,0.79d0]; x(:,7)=[0.78d0,0.80d0]; x(:,8)=[0.83d0,0.81d0]
    x(:,9)=[0.15d0,0.85d0]; x(:,10)=[0.14d0,0.88d0]; x(:,11)=[0.17d0,0.86d0]; x(:,12)=[0.16d0,0.83d0]
    ! interleave the order so that labels are not trivially sorted
    x = x(:,[1,5,9,2,6,10,3,7,11,4,8,12])
    S = calculate_similarity_matrix(x)
    labels = NN_clustering(S,ncl)
    write(*,'(A)',advance='no') '"knn_points": ['
    do i=1,12
        write(*,'(ES24.16,A,ES24.16)',advance='no') x(1,i), ',', x(2,i)
        if (i<12) write(*,'(A)',advance='no') ','
    end do
    write(*,'(A)') '],'
    write(*,'(A,I0,A)') '"knn_nclusters": ', ncl, ','
    write(*,'(A)',advance='no') '"knn_labels": ['
    do i=1,12
        write(*,'(I0)',advance='no') labels(i)
        if (i<12) write(*,'(A)',advance='no') ','
    end do
    write(*,'(A)') '],'
    knn = compute_knn(S,4)
    write(*,'(A)',advance='no') '"knn4": ['
    do i=1,12
        do j=1,4
            write(*,'(I0)',advance='no') knn(j,i)
            if (.not.(i==12.and.j==4)) write(*,'(A)',advance='no') ','
        end do
    end do
    write(*,'(A)') '],'
    write(*,'(A,ES24.16,A)') '"similarity_1_2": ', S(1,2), ','
    lab2 = relabel([7,7,3,9,3,7,1,9],nl)
    write(*,'(A)',advance='no') '"relabel": ['
    do i=1,8
        write(*,'(I0)',advance='no') lab2(i)
        if (i<8) write(*,'(A)',advance='no') ','
    end do
    write(*,'(A)') ']'
    write(*,'(A)') '}'
end program ref_units
